// Issue-rate calibration for the integer VALU instructions the Myers kernels are made of (gfx950).
// Each kernel runs a long unrolled stream of one instruction kind on 8 independent register chains;
// the grid puts 8 wavefronts on every SIMD.  Prints shader cycles per wavefront-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(ASM) \
    for (int it = 0; it < iters; it++) { REP8(REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)) }

template <int KIND> __global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint64_t *cycles)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b0 = a0 ^ 0x55, b1 = a0 + 77;
    uint64_t t0 = __builtin_readcyclecounter();
    // every asm statement = 4 instructions (one per chain)
    if (KIND == 0) BODY("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %5")
    if (KIND == 1) BODY("v_bitop3_b32 %0, %0, %4, %5 bitop3:0xf1\n v_bitop3_b32 %1, %1, %5, %4 bitop3:0xf1\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0xbe\n v_bitop3_b32 %3, %3, %5, %4 bitop3:0xbe")
    if (KIND == 2) BODY("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %4, 31\n v_alignbit_b32 %3, %3, %5, 31")
    if (KIND == 3) BODY("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %5\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %5")
    if (KIND == 4) BODY("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0")
    if (KIND == 5) BODY("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %5, vcc")
    if (KIND == 6) BODY("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %5, %4")
    if (KIND == 7) BODY("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5")
    if (KIND == 8) BODY("v_lshrrev_b32 %0, 31, %0\n v_lshrrev_b32 %1, 31, %1\n v_lshrrev_b32 %2, 31, %2\n v_lshrrev_b32 %3, 31, %3")
    if (KIND == 9) BODY("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %5, %4\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %5, %4")
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// KIND 4 needs 64-bit register pairs: its own kernel
__global__ void __launch_bounds__(256) k64(uint64_t *out, int iters, uint64_t *cycles)
{
    uint64_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b0 = a0 ^ 0x55, b1 = a0 + 77;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        REP8(REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %5\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));))
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
__global__ void __launch_bounds__(256) kshl64(uint64_t *out, int iters, uint64_t *cycles)
{
    uint64_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        REP8(REP8(asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));))
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <class F> void run(const char *name, F launch, int waves_per_simd, int n_cu)
{
    const int iters = 2000;                       // x 64 statements x 4 instructions
    const double inst_per_wave = (double)iters * 64 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uint64_t *cyc; hipMalloc(&cyc, 8);
    launch(1, cyc);                                // warm-up
    hipDeviceSynchronize();
    const int blocks = n_cu * waves_per_simd;      // 256 threads = 4 wavefronts = one per SIMD
    hipEventRecord(e0);
    launch(blocks, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_simd = inst_per_wave * waves_per_simd;   // instructions each SIMD issued
    printf("%-16s waves/SIMD %d  %.3f ms  wall-clock cycles@2.4GHz per instr per SIMD %.2f   wave0 counter ticks %.0f (%.3f per instr)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / per_simd, (double)h, (double)h / inst_per_wave);
    hipFree(cyc);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    printf("%s  CUs %d  clock %d kHz\n", p.gcnArchName, n_cu, p.clockRate);
    uint32_t *out; hipMalloc(&out, (size_t)n_cu * 8 * 256 * 8);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
#define RUN(K, NAME) run(NAME, [&](int blocks, uint64_t *cyc) { hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); }, wps, n_cu)
        RUN(0, "v_and_b32"); RUN(1, "v_bitop3_b32"); RUN(2, "v_alignbit_b32"); RUN(3, "v_lshl_or_b32");
        RUN(5, "v_add_co/addc"); RUN(6, "v_add3_u32"); RUN(7, "v_add_u32"); RUN(8, "v_lshrrev_b32"); RUN(9, "v_or3_b32");
        run("v_lshl_add_u64", [&](int blocks, uint64_t *cyc) { hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, (uint64_t *)out, iters, cyc); }, wps, n_cu);
        run("v_lshlrev_b64", [&](int blocks, uint64_t *cyc) { hipLaunchKernelGGL(kshl64, dim3(blocks), dim3(256), 0, 0, (uint64_t *)out, iters, cyc); }, wps, n_cu);
    }
    return 0;
}
