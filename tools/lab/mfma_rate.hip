// Microbenchmark: sustained rate of v_mfma_f32_16x16x32_bf16 with NACC independent accumulators,
// 8 waves per CU (2 per SIMD), every CU busy.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC * 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, int threads, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * (threads / 64) * iters * 64.0;
    const double flops = mfmas * 16384.0;
    printf("NACC=%d blocks=%d threads=%d: %.3f ms, %.1f TFLOP/s, %.2f ns per MFMA per SIMD-slot\n", NACC, blocks, threads, ms,
           flops / ms / 1e9, ms * 1e6 / (mfmas / (256.0 * 4)));
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 512 * sizeof(float));
    run<4>(256, 512, d, 4096);
    run<8>(256, 512, d, 4096);
    run<4>(256, 256, d, 4096);
    run<4>(512, 512, d, 2048);
    run<16>(256, 512, d, 4096);
    return 0;
}
