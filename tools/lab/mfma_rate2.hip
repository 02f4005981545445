// Microbenchmark 2: the W-phase MFMA pattern (4 accumulators, 2 A fragments x 8 B fragments per 16-MFMA
// batch) with CONSTANT vs RANDOM operand data: is the sustained rate data (power) dependent?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
union F8 { uint4 u; bf16x8 v; };
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, float* out, int iters) {
    F8 a[4], b[16];
    for (int i = 0; i < 4; ++i) a[i].u = src[(threadIdx.x * 20 + i) & 4095];
    for (int i = 0; i < 16; ++i) b[i].u = src[(threadIdx.x * 20 + 4 + i) & 4095];
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(s + t) & 3].v, b[4 * s + j].v, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
    uint4* src; hipMalloc(&src, 4096 * sizeof(uint4));
    unsigned* h = (unsigned*)malloc(4096 * 16);
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < 4096 * 4; ++i) {
            if (mode == 0) h[i] = 0x3f803f80u;                                   // all ones (bf16 1.0)
            else if (mode == 1) { unsigned r = rand(); h[i] = (0x3f80u | (r & 0x7f)) | ((0x3f80u | ((r >> 8) & 0x7f)) << 16); }  // random mantissa in [1,2)
            else { unsigned r1 = rand(), r2 = rand(); h[i] = ((0x3c00u + (r1 & 0x7ff)) | ((r1 >> 15) & 1) << 15) | (((0x3c00u + (r2 & 0x7ff)) | ((r2 >> 15) & 1) << 15) << 16); }  // random sign/exponent/mantissa
        }
        hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, src, d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, src, d, 2048);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = 256.0 * 8 * 2048 * 64.0;
        printf("data mode %d: %.3f ms, %.1f TFLOP/s, %.2f ns per MFMA per SIMD\n", mode, ms, mfmas * 16384.0 / ms / 1e9,
               ms * 1e6 / (mfmas / 1024.0));
    }
    return 0;
}
