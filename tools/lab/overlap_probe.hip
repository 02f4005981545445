// Premise check for running AO-ADMM's one-workgroup `prepare` BESIDE a V-sized product (r4):
// kernel A stands for the product (512 threads, 160 KiB of LDS: one block per CU) and spins for `ta` us per block,
// kernel B stands for prepare (576 threads, 72 KiB) and spins for `tb` us, launched on a second stream.
//   mode 0: A alone (256 blocks)            mode 1: A (256 blocks) then B, one stream
//   mode 2: A with 256 blocks, B on stream 2 launched FIRST           (B takes a CU, one A block starts late)
//   mode 3: A with 248 working blocks (blocks 248..255 of a 256-block grid return at once), B on stream 2 launched first
//   mode 4: as 3, B launched AFTER A
//   mode 5: A with 255 working blocks (block 255 returns at once), B first
//   mode 6: A as a grid of 248 blocks, B first
//   modes 7 / 8 / 9: ONE stream, 248 blocks: B then A launched with hipExtAnyOrderLaunch / A then B with it / both ordinary
// Prints the wall time of the pair (events on a third "join" pattern: stream 1 waits for stream 2's event).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void spin_a(long long ticks, int working, unsigned* xcc_of, unsigned long long* t_start)
{
    extern __shared__ unsigned char lds[];
    if ((int)blockIdx.x >= working) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc_of[blockIdx.x] = xcc & 15;
        t_start[blockIdx.x] = t0;
    }
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(576) void spin_b(long long ticks, unsigned* xcc_of, unsigned long long* t_start)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc_of[0] = xcc & 15;
        t_start[0] = t0;
    }
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv)
{
    const double ta = argc > 1 ? atof(argv[1]) : 100.0, tb = argc > 2 ? atof(argv[2]) : 38.0;
    const long long ka = (long long)(ta * 100.0), kb = (long long)(tb * 100.0);      // s_memrealtime: 100 MHz
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    unsigned *xa, *xb; unsigned long long *sa, *sb;
    CK(hipMalloc(&xa, 256 * 4)); CK(hipMalloc(&xb, 4)); CK(hipMalloc(&sa, 256 * 8)); CK(hipMalloc(&sb, 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_a), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_b), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    for (int mode = 0; mode <= 9; ++mode) {
        std::vector<float> ms;
        unsigned hxa[256], hxb = 99; unsigned long long hsa[256], hsb = 0;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipMemsetAsync(xa, 0xff, 256 * 4, s1));
            CK(hipEventRecord(e0, s1));
            const int grid = mode >= 6 ? 248 : 256;
            const int working = (mode == 3 || mode == 4) ? 248 : mode == 5 ? 255 : 256;
            if (mode == 0) {
                hipLaunchKernelGGL(spin_a, dim3(grid), dim3(512), 160 * 1024, s1, ka, working, xa, sa);
            } else if (mode == 1) {
                hipLaunchKernelGGL(spin_a, dim3(grid), dim3(512), 160 * 1024, s1, ka, working, xa, sa);
                hipLaunchKernelGGL(spin_b, dim3(1), dim3(576), 72 * 1024, s1, kb, xb, sb);
            } else if (mode == 7) {          // one stream: B, then A without the barrier bit (hipExtAnyOrderLaunch)
                hipLaunchKernelGGL(spin_b, dim3(1), dim3(576), 72 * 1024, s1, kb, xb, sb);
                hipExtLaunchKernelGGL(spin_a, dim3(grid), dim3(512), 160 * 1024, s1, nullptr, nullptr, hipExtAnyOrderLaunch, ka, working, xa, sa);
            } else if (mode == 8) {          // one stream: A, then B without the barrier bit
                hipLaunchKernelGGL(spin_a, dim3(grid), dim3(512), 160 * 1024, s1, ka, working, xa, sa);
                hipExtLaunchKernelGGL(spin_b, dim3(1), dim3(576), 72 * 1024, s1, nullptr, nullptr, hipExtAnyOrderLaunch, kb, xb, sb);
            } else if (mode == 9) {          // one stream, 248 blocks, both ordinary launches (reference for 7 / 8)
                hipLaunchKernelGGL(spin_b, dim3(1), dim3(576), 72 * 1024, s1, kb, xb, sb);
                hipLaunchKernelGGL(spin_a, dim3(grid), dim3(512), 160 * 1024, s1, ka, working, xa, sa);
            } else {
                CK(hipEventRecord(fork, s1));
                CK(hipStreamWaitEvent(s2, fork, 0));
                if (mode != 4) hipLaunchKernelGGL(spin_b, dim3(1), dim3(576), 72 * 1024, s2, kb, xb, sb);
                hipLaunchKernelGGL(spin_a, dim3(grid), dim3(512), 160 * 1024, s1, ka, working, xa, sa);
                if (mode == 4) hipLaunchKernelGGL(spin_b, dim3(1), dim3(576), 72 * 1024, s2, kb, xb, sb);
                CK(hipEventRecord(join, s2));
                CK(hipStreamWaitEvent(s1, join, 0));
            }
            CK(hipEventRecord(e1, s1));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(s2));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep >= 2) ms.push_back(t * 1000.f);
            CK(hipMemcpy(hxa, xa, sizeof(hxa), hipMemcpyDeviceToHost));
            CK(hipMemcpy(hsa, sa, sizeof(hsa), hipMemcpyDeviceToHost));
            if (mode) { CK(hipMemcpy(&hxb, xb, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hsb, sb, 8, hipMemcpyDeviceToHost)); }
        }
        std::sort(ms.begin(), ms.end());
        int per_xcc[16] = {0};
        unsigned long long first = ~0ull, last = 0;
        for (int b = 0; b < 256; ++b) if (hxa[b] < 16) { per_xcc[hxa[b]]++; first = std::min(first, hsa[b]); last = std::max(last, hsa[b]); }
        printf("mode %d: pair %.1f us (min %.1f max %.1f)  A blocks per XCC:", mode, ms[ms.size() / 2], ms.front(), ms.back());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("  last A start - first A start %.1f us", (double)(last - first) / 100.0);
        if (mode) printf("  B on XCC %u, B start - first A start %.1f us", hxb, ((double)hsb - (double)first) / 100.0);
        printf("  block0 XCC %u block1 XCC %u\n", hxa[0], hxa[1]);
    }
    return 0;
}
