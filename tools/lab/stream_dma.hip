// The same 512 MiB read through LDS-DMA (global_load_lds_dwordx4 nt) with NO consumer: every wave streams its own pieces
// into a private LDS ring, D pieces of 4 KiB deep.  Is the DMA path slower than loads into registers (7.0 TB/s)?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int D, bool NT>
__global__ __launch_bounds__(512) void dma(const char* __restrict__ p, size_t bytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per_block = bytes / gridDim.x, per_wave = per_block / 8;
    unsigned long long base = (unsigned long long)p + per_block * blockIdx.x + per_wave * wave;
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm + wave * (D * 4096);
    const unsigned off = lane * 16;
    const int steps = (int)(per_wave / 4096);
    int slot = 0;
    for (int s = 0; s < steps; ++s) {
        const unsigned dst = ldsbase + slot * 4096;
        unsigned keep;
        if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
                     "global_load_lds_dwordx4 %3, %1 offset:1024 nt\n\tglobal_load_lds_dwordx4 %3, %1 offset:2048 nt\n\t"
                     "global_load_lds_dwordx4 %3, %1 offset:3072 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(base), "s"(dst), "v"(off) : "memory");
        else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                     "global_load_lds_dwordx4 %3, %1 offset:1024\n\tglobal_load_lds_dwordx4 %3, %1 offset:2048\n\t"
                     "global_load_lds_dwordx4 %3, %1 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(base), "s"(dst), "v"(off) : "memory");
        base += 4096;
        slot = (slot == D - 1) ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (D - 1)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sm[threadIdx.x] == 77 && out) out[0] = 1;
}
template <typename K> void run(const char* name, K kern, int grid, size_t shm, const char* d, size_t bytes, unsigned* o) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, 0, d, bytes, o);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shm, 0, d, bytes, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-22s grid %4d: %7.1f us  %5.2f TB/s\n", name, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = 512ull << 20;
    char* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    run("dma nt depth 2 (64K)", dma<2, true>, 256, 8 * 2 * 4096, d, bytes, o);
    run("dma nt depth 3 (96K)", dma<3, true>, 256, 8 * 3 * 4096, d, bytes, o);
    run("dma nt depth 4 (128K)", dma<4, true>, 256, 8 * 4 * 4096, d, bytes, o);
    run("dma nt depth 5 (160K)", dma<5, true>, 256, 8 * 5 * 4096, d, bytes, o);
    run("dma    depth 4 (128K)", dma<4, false>, 256, 8 * 4 * 4096, d, bytes, o);
    run("dma nt depth 2, 2 blk/CU", dma<2, true>, 512, 8 * 2 * 4096, d, bytes, o);
    return 0;
}
