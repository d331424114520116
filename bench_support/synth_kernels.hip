// synth_kernels.hip -- seeded synthetic workload generators for bench.py / large tests
// (NOT part of the product ABI; built into real_amd/libreal_synth.so).
// Reproduces the distributions of the reference's time(0)-seeded tools:
//   randstr.cpp:27-53  i.i.d. uniform ACGT genome
//   genpat.cpp:96-157  reads copied from uniform start positions, strand flip p=0.5,
//                      per-base substitution to a different base with probability errprob,
//                      FASTQ quality 'D' (unchanged) / '*' (mutated)
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_genome_kernel(uint8_t *sym, uint64_t n, uint64_t seed)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // 32 symbols per thread
    uint64_t base = t * 32;
    if (base >= n) return;
    uint64_t s = seed * 0xD1342543DE82EF95ull + t;
    uint64_t r = splitmix64(s);
    for (int i = 0; i < 32 && base + i < n; ++i) sym[base + i] = (uint8_t)((r >> (2 * i)) & 3);
}

__global__ void synth_positions_kernel(int64_t *pos, uint64_t n_reads, uint64_t numpos, uint64_t seed)
{
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t s = seed * 0xA24BAED4963EE407ull + r;
    // 128-bit multiply-shift: unbiased enough for numpos << 2^64
    pos[r] = (int64_t)__umul64hi(splitmix64(s), numpos);
}

__global__ void synth_reads_kernel(const uint8_t *__restrict__ sym, const int64_t *__restrict__ pos, uint64_t n_reads,
                                   uint32_t patl, uint32_t err_thresh /* errprob * 2^32 */, uint64_t seed,
                                   uint8_t *__restrict__ bases, uint8_t *__restrict__ qual, uint8_t *__restrict__ inv_out)
{
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t s = seed * 0x9FB21C651E98DF25ull + r;
    const bool inv = splitmix64(s) & 1;
    const uint8_t *g = sym + pos[r];
    uint8_t *b = bases + r * (uint64_t)patl, *q = qual + r * (uint64_t)patl;
    for (uint32_t i = 0; i < patl; ++i) {
        uint8_t c = inv ? (uint8_t)(3 - g[patl - 1 - i]) : g[i];
        uint64_t x = splitmix64(s);
        bool mut = (uint32_t)x < err_thresh && c < 4;
        if (mut) c = (uint8_t)((c + 1 + ((x >> 32) % 3)) & 3);
        b[i] = c;
        q[i] = mut ? 9 : 35; // '*'-33 : 'D'-33
    }
    if (inv_out) inv_out[r] = inv;
}

extern "C" int real_synth_genome(uint8_t *d_sym, uint64_t n, uint64_t seed)
{
    uint64_t t = (n + 31) / 32;
    if (t) hipLaunchKernelGGL(synth_genome_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, 0, d_sym, n, seed);
    return (int)hipDeviceSynchronize();
}
extern "C" int real_synth_positions(int64_t *d_pos, uint64_t n_reads, uint64_t numpos, uint64_t seed)
{
    if (n_reads) hipLaunchKernelGGL(synth_positions_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, 0, d_pos, n_reads, numpos, seed);
    return (int)hipDeviceSynchronize();
}
extern "C" int real_synth_reads(const uint8_t *d_sym, const int64_t *d_pos, uint64_t n_reads, uint32_t patl, double errprob,
                                uint64_t seed, uint8_t *d_bases, uint8_t *d_qual, uint8_t *d_inv)
{
    uint32_t th = (uint32_t)(errprob * 4294967296.0);
    if (n_reads) hipLaunchKernelGGL(synth_reads_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, 0, d_sym, d_pos, n_reads, patl, th, seed, d_bases, d_qual, d_inv);
    return (int)hipDeviceSynchronize();
}
