// randread6.hip -- like randread, but every lane reads one 8-byte word from each of NB separate
// allocations per round (the bucket-table phase of the matcher: six tables, one request each).
//   usage: randread6 <gb_per_buffer> <rounds> <nbuf 1..6> <waves_per_simd>
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
struct Bufs { const uint64_t *p[6]; };
template <int NB>
__global__ void k(Bufs b, uint64_t nwords, int rounds, uint64_t *out)
{
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1), acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint64_t v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = b.p[i][__umul64hi(mix(s + i * 0x632BE59BD9B4E019ull), nwords)];
#pragma unroll
        for (int i = 0; i < NB; ++i) acc += v[i];
        s = mix(s ^ acc);
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main(int argc, char **argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 8; int rounds = argc > 2 ? atoi(argv[2]) : 256; int nb = argc > 3 ? atoi(argv[3]) : 6; int wps = argc > 4 ? atoi(argv[4]) : 3;
    uint64_t nwords = (uint64_t)(gb * (1ull << 30)) / 8; Bufs b; uint64_t *out;
    for (int i = 0; i < 6; ++i) { void *p = nullptr; if (i < nb) { if (hipMalloc(&p, nwords * 8) != hipSuccess) { printf("alloc failed\n"); return 1; } (void)hipMemset(p, 1, nwords * 8); } b.p[i] = (const uint64_t *)p; }
    (void)hipMalloc(&out, 16);
    int blocks = 256 * wps; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        switch (nb) { case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, b, nwords, rounds, out); break;
                      case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, b, nwords, rounds, out); break;
                      case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, b, nwords, rounds, out); break;
                      default: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, b, nwords, rounds, out); break; }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        double acc = (double)blocks * 256 * rounds * (nb > 3 ? 6 : nb);
        if (it == 2) printf("nbuf=%d x %.1f GiB rounds=%d waves/simd=%d : %.3f ms, %.2f G reads/s, round trip %.2f us\n", nb, gb, rounds, wps, ms, acc / ms / 1e6, ms * 1e3 / rounds);
    }
    return 0;
}
