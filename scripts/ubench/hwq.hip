// How many kernels of one process run side by side: N streams, one long single-wavefront kernel each.
//   GPU_MAX_HW_QUEUES=8 ./hwq      (or ./hwq set  -> setenv() inside the process before the first HIP call)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void spin(long long cycles, int *out)
{
    long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (out) *out = 1;
}
int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "set")) setenv("GPU_MAX_HW_QUEUES", "8", 1);
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", e ? e : "(unset)");
    for (int n : {1, 2, 4, 5, 6, 8, 12}) {
        hipStream_t st[16];
        for (int i = 0; i < n; i++) (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[0], 1000, nullptr);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, st[0]);
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 100000000LL / 50, nullptr); // ~20 ms at 100 MHz
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e1, st[0]);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("streams %2d: %.2f ms\n", n, ms);
        for (int i = 0; i < n; i++) (void)hipStreamDestroy(st[i]);
    }
    return 0;
}
