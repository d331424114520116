// wallclock.hip -- the rate of s_memrealtime (the stamp of the matcher's phase-timing build) against HIP events
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *out, int n)
{
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[0] = t1 - t0;
}
int main()
{
    int rate = 0;
    (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    unsigned long long *d, h = 0;
    (void)hipMalloc(&d, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 20000);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("wall clock rate attribute %d kHz; kernel %.3f ms by events, %llu ticks => %.1f ticks/us\n", rate, ms, h, h / (ms * 1e3));
    }
    return 0;
}
