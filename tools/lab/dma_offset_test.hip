// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination too?  (tools/lab: hardware probe)
//   hipcc --offload-arch=gfx950 -O2 tools/lab/dma_offset_test.hip -o /tmp/dma_offset_test && /tmp/dma_offset_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* p, float* out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = -1.f;
    __syncthreads();
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) float*)sm;
    const unsigned long long base = (unsigned long long)p;
    const unsigned o0 = threadIdx.x * 16, o1 = 4096 + threadIdx.x * 16 - 1024;     // piece 1: 1024 floats further, minus the instruction offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
                 "global_load_lds_dwordx4 %4, %1 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(dst), "v"(o0), "v"(o1) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = sm[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, o);
    std::vector<float> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    printf("lds[0..3]    = %g %g %g %g   (piece 0 expects 0 1 2 3)\n", r[0], r[1], r[2], r[3]);
    printf("lds[256..259] = %g %g %g %g   (piece 1 lands here iff the offset moves the LDS address: expects 1024 1025 1026 1027)\n", r[256], r[257], r[258], r[259]);
    printf("lds[512] = %g\n", r[512]);
    return 0;
}
