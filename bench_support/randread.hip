// randread.hip -- micro-benchmark: what random small-read rate does MI355X HBM sustain?
// The read-matching kernel is a dependent random-access workload (bucket table -> entries ->
// text), so its practical roof is the random 64-byte-sector rate of the memory system, not the
// 8 TB/s streaming peak.  Each lane issues ROUNDS batches of MLP independent 8-byte loads at
// hashed addresses inside a buffer of `gb` GiB; the next batch depends on the previous one.
//   usage: randread <gb> <rounds> <mlp:1|2|4|8> <waves_per_simd:1..8>
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
__global__ void randread_kernel(const uint64_t *__restrict__ buf, uint64_t nwords, int rounds, uint64_t *out, int active)
{
    const uint64_t t_start = __builtin_amdgcn_s_memtime();
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint64_t acc = 0;
    if ((int)(threadIdx.x & 63) >= active) return; // only `active` lanes of every wave issue loads
    for (int r = 0; r < rounds; ++r) {
        uint64_t v[MLP];
#pragma unroll
        for (int k = 0; k < MLP; ++k) {
            uint64_t h = mix(s + k * 0x632BE59BD9B4E019ull);
            v[k] = buf[__umul64hi(h, nwords)];
        }
#pragma unroll
        for (int k = 0; k < MLP; ++k) acc += v[k];
        s = mix(s ^ acc); // next addresses depend on the data: a dependent chain like bucket -> entry -> text
    }
    if (acc == 0x1234567) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = __builtin_amdgcn_s_memtime() - t_start;
}

int main(int argc, char **argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 64;
    int rounds = argc > 2 ? atoi(argv[2]) : 64;
    int mlp = argc > 3 ? atoi(argv[3]) : 1;
    int wps = argc > 4 ? atoi(argv[4]) : 8;
    int active = argc > 6 ? atoi(argv[6]) : 64;
    uint64_t nwords = (uint64_t)(gb * (1ull << 30)) / 8;
    uint64_t *buf, *out;
    if (hipMalloc(&buf, nwords * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 16);
    hipMemset(buf, 1, nwords * 8);
    int blocks = argc > 5 ? atoi(argv[5]) : 256 * wps; // default: 256 CUs x wps blocks of 256 threads = wps waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        switch (mlp) {
        case 1: hipLaunchKernelGGL(randread_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, nwords, rounds, out, active); break;
        case 2: hipLaunchKernelGGL(randread_kernel<2>, dim3(blocks), dim3(256), 0, 0, buf, nwords, rounds, out, active); break;
        case 4: hipLaunchKernelGGL(randread_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, nwords, rounds, out, active); break;
        default: hipLaunchKernelGGL(randread_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, nwords, rounds, out, active); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double acc = (double)blocks * 4 * active * rounds * mlp;
        uint64_t hc[2] = {0, 0};
        hipMemcpy(hc, out, 16, hipMemcpyDeviceToHost);
        if (it == 2)
            printf("[block 0 ran %.0f kcycles in %.3f ms => shader clock >= %.2f GHz] ", hc[1] / 1e3, ms, hc[1] / (ms * 1e6));
        if (it == 2)
            printf("active=%d gb=%.0f rounds=%d mlp=%d waves/simd=%d : %.3f ms, %.2f G reads/s, %.2f TB/s at 64 B/sector, round trip %.2f us\n", active, gb, rounds, mlp,
                   wps, ms, acc / ms / 1e6, acc * 64 / ms / 1e9, ms * 1e3 / rounds);
    }
    return 0;
}
