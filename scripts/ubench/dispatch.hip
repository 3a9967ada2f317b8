// When do kernels on different streams START when each has more workgroups than the chip holds at once?
// N streams, one kernel each (B workgroups of 64 threads, L bytes of LDS, every workgroup busy for about U microseconds);
// each kernel notes the time its first workgroup started and its last one ended (constant-rate wall clock, 100 MHz).
//   GPU_MAX_HW_QUEUES=16 ./dispatch [streams=8] [blocks=20000] [lds_bytes=16384] [us=20]
//   GPU_MAX_HW_QUEUES=16 ./dispatch spec blocks:us:lds[:stream] ...     one kernel per argument, in launch order
//                                                                         (stream = index of the stream it goes to, default its own)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void __launch_bounds__(64) busy(unsigned long long ticks, unsigned long long *first, unsigned long long *last)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(first, t0);
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) lds[threadIdx.x] = 1; // keeps the loop and the LDS allocation alive
    if (threadIdx.x == 0) atomicMax(last, wall_clock64());
}
static int run_spec(int argc, char **argv)
{
    struct K { int blocks, us, lds, stream; };
    std::vector<K> ks;
    for (int i = 2; i < argc; i++) {
        K k{0, 0, 0, i - 2};
        const int got = sscanf(argv[i], "%d:%d:%d:%d", &k.blocks, &k.us, &k.lds, &k.stream);
        if (got < 3 || k.blocks < 1 || k.blocks > 1000000 || k.us < 1 || k.us > 20000 || k.lds < 0 || k.lds > 65536 || k.stream < 0 || k.stream >= 32) { printf("bad spec %s\n", argv[i]); return 1; }
        ks.push_back(k);
    }
    const int n = (int)ks.size();
    if (n < 1 || n > 32) return 1;
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s kernels=%d\n", e ? e : "(unset)", n);
    std::vector<hipStream_t> st(32);
    for (int i = 0; i < 32; i++) if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) return 2;
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(unsigned long long) * 2 * n) != hipSuccess) return 2;
    std::vector<unsigned long long> h(2 * n);
    for (int rep = 0; rep < 2; rep++) {
        for (int i = 0; i < n; i++) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        (void)hipMemcpy(d, h.data(), sizeof(unsigned long long) * 2 * n, hipMemcpyHostToDevice);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < n; i++)
            hipLaunchKernelGGL(busy, dim3(ks[i].blocks), dim3(64), ks[i].lds, st[ks[i].stream], (unsigned long long)ks[i].us * 100, d + 2 * i, d + 2 * i + 1);
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < n; i++) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
        if (rep == 0) continue;
        for (int i = 0; i < n; i++)
            printf("  kernel %2d (%7d blocks x %5d us, lds %5d, stream %2d): first workgroup at %8.3f ms, last done at %8.3f ms\n", i, ks[i].blocks, ks[i].us,
                   ks[i].lds, ks[i].stream, (h[2 * i] - t0) / 1e5, (h[2 * i + 1] - t0) / 1e5);
    }
    for (int i = 0; i < 32; i++) (void)hipStreamDestroy(st[i]);
    (void)hipFree(d);
    return 0;
}
int main(int argc, char **argv)
{
    if (argc > 2 && !strcmp(argv[1], "spec")) return run_spec(argc, argv);
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = argc > 2 ? atoi(argv[2]) : 20000;
    const int lds = argc > 3 ? atoi(argv[3]) : 16384;
    const int us = argc > 4 ? atoi(argv[4]) : 20;
    if (n < 1 || n > 32 || blocks < 1 || blocks > 1000000 || lds < 0 || lds > 65536 || us < 1 || us > 1000) { printf("bad arguments\n"); return 1; }
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s streams=%d blocks=%d lds=%d us=%d\n", e ? e : "(unset)", n, blocks, lds, us);
    std::vector<hipStream_t> st(n);
    for (int i = 0; i < n; i++) if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) return 2;
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(unsigned long long) * 2 * n) != hipSuccess) return 2;
    std::vector<unsigned long long> h(2 * n);
    for (int rep = 0; rep < 2; rep++) {
        for (int i = 0; i < n; i++) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        (void)hipMemcpy(d, h.data(), sizeof(unsigned long long) * 2 * n, hipMemcpyHostToDevice);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(busy, dim3(blocks), dim3(64), lds, st[i], (unsigned long long)us * 100, d + 2 * i, d + 2 * i + 1);
        if (hipDeviceSynchronize() != hipSuccess) return 3;
        (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < n; i++) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
        if (rep == 0) continue; // first round warms up
        for (int i = 0; i < n; i++) printf("  stream %2d: first workgroup at %8.3f ms, last done at %8.3f ms\n", i, (h[2 * i] - t0) / 1e5, (h[2 * i + 1] - t0) / 1e5);
    }
    for (int i = 0; i < n; i++) (void)hipStreamDestroy(st[i]);
    (void)hipFree(d);
    return 0;
}
