// r5 probe for the "cross terms on the fp8 pipe" idea (LAB_NOTES R5): does this toolchain have the block-scaled fp8 MFMA of gfx950,
// what is its operand layout, what does the scale operand do, and what does a 2 bf16 + 4 fp8 term mix sustain against 6 bf16 terms
// on RANDOM operands (the chip is power-limited: the nominal 2x of the fp8 pipe need not survive)?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_fp8_probe tools/lab/mfma_fp8_probe.hip && ./mfma_fp8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// ---- 1. layout / scale semantics: one instruction, small-integer data ----------------------------------------------------------
__global__ void one_mfma(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, float* __restrict__ C, int scale_a, int scale_b)
{
    const int lane = threadIdx.x;
    i32x8 a, b;
    // assumed layout: lane l holds row (col) l % 32, k = 32 (l / 32) .. + 31, four bytes per register in k order
    for (int r = 0; r < 8; ++r) {
        unsigned wa = 0, wb = 0;
        for (int e = 0; e < 4; ++e) {
            const int k = 32 * (lane >> 5) + 4 * r + e;
            wa |= (unsigned)A[(lane & 31) * 64 + k] << (8 * e);
            wb |= (unsigned)B[(lane & 31) * 64 + k] << (8 * e);
        }
        a[r] = (int)wa; b[r] = (int)wb;
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);      // cbsz = blgp = 0: both e4m3
    for (int i = 0; i < 16; ++i) C[i * 64 + lane] = c[i];
}

static unsigned char e4m3_of_small_int(int v) {       // exact for |v| <= 15
    if (v == 0) return 0;
    const int s = v < 0; int a = s ? -v : v;
    int e = 0; while ((a >> (e + 1)) != 0) ++e;        // a in [2^e, 2^(e+1))
    const int mant = ((a << 3) >> e) & 7;              // three bits below the leading one (exact for a < 16)
    return (unsigned char)((s << 7) | ((e + 7) << 3) | mant);
}

// ---- 2. sustained rate: 6 bf16 terms against 2 bf16 + 4 fp8 terms per K = 64 ----------------------------------------------------
union B8 { uint4 u; bf16x8 v; };
template <int MODE> __global__ __launch_bounds__(512) void rate(const uint4* __restrict__ src, float* out, int iters)
{
    B8 a[4], b[4];
    i32x8 fa[2], fb[2];
    for (int i = 0; i < 4; ++i) { a[i].u = src[(threadIdx.x * 24 + i) & 4095]; b[i].u = src[(threadIdx.x * 24 + 4 + i) & 4095]; }
    for (int i = 0; i < 2; ++i) {
        const uint4 p = src[(threadIdx.x * 24 + 8 + 2 * i) & 4095], q = src[(threadIdx.x * 24 + 9 + 2 * i) & 4095];
        // (fp8 bit patterns with the exponent kept away from NaN: clear bit 6 of every byte)
        fa[i] = (i32x8){(int)(p.x & 0xbfbfbfbf), (int)(p.y & 0xbfbfbfbf), (int)(p.z & 0xbfbfbfbf), (int)(p.w & 0xbfbfbfbf),
                        (int)(q.x & 0xbfbfbfbf), (int)(q.y & 0xbfbfbfbf), (int)(q.z & 0xbfbfbfbf), (int)(q.w & 0xbfbfbfbf)};
        fb[i] = (i32x8){(int)(q.x & 0xbfbfbfbf), (int)(p.y & 0xbfbfbfbf), (int)(q.z & 0xbfbfbfbf), (int)(p.w & 0xbfbfbfbf),
                        (int)(p.x & 0xbfbfbfbf), (int)(q.y & 0xbfbfbfbf), (int)(p.z & 0xbfbfbfbf), (int)(q.w & 0xbfbfbfbf)};
    }
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {                  // two accumulator tiles; per tile and K = 64: `terms` products
            if (MODE == 0) {                           // six bf16 terms = 24 instructions of K = 16
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + t) & 3].v, b[(s + j) & 3].v, acc[j], 0, 0, 0);
            } else if (MODE == 1) {                    // two bf16 terms + four fp8 terms (one K = 64 instruction each)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + t) & 3].v, b[(s + j) & 3].v, acc[j], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[t & 1], fb[(t >> 1) & 1], acc[j], 0, 0, 0, 127, 0, 127);
            } else {                                   // fp8 only: 6 terms
#pragma unroll
                for (int t = 0; t < 6; ++t) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[t & 1], fb[(t >> 1) & 1], acc[j], 0, 0, 0, 127, 0, 127);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> static void time_rate(const uint4* src, float* d, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(512), 0, 0, src, d, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(512), 0, 0, src, d, 4000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double units = 256.0 * 8 * 4000 * 2;         // (wave, iteration, tile): six 32 x 32 x 64 products each
    printf("%-44s %.3f ms  = %.1f ns per (tile, K = 64, six terms) per wave; %.0f TFLOP/s in product terms\n", what, ms, ms * 1e6 / (units / 8.0) / 256.0 * 256.0 / 1.0 * 0 + ms * 1e6 / (4000.0 * 2),
           units * 6.0 * 2.0 * 32 * 32 * 64 / ms / 1e9);
}

int main() {
    // 1. layout
    unsigned char hA[32 * 64], hB[32 * 64];
    int iA[32 * 64], iB[32 * 64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { iA[i] = rand() % 15 - 7; iB[i] = rand() % 15 - 7; hA[i] = e4m3_of_small_int(iA[i]); hB[i] = e4m3_of_small_int(iB[i]); }
    unsigned char *dA, *dB; float* dC; float hC[16 * 64];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 3; ++pass) {
        const int sa = pass == 1 ? 128 : 127, sb = pass == 2 ? 125 : 127;          // E8M0: 2^(s - 127)
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb);
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // expected: C[row][col] = sum_k A[row][k] B[col][k]; accumulator register i of lane l: col = l % 32, row = (i % 4) + 8 (i / 4) + 4 (l / 32)
        double worst = 0.0; int bad = 0;
        const double f = std::ldexp(1.0, (sa - 127) + (sb - 127));
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
                double want = 0.0;
                for (int k = 0; k < 64; ++k) want += (double)iA[row * 64 + k] * iB[col * 64 + k];
                want *= f;
                const double d = std::fabs(hC[i * 64 + l] - want);
                if (d > worst) worst = d;
                if (d > 1e-3) ++bad;
            }
        printf("layout check, scale bytes (%d, %d): worst |C - expected| = %g, %d of 1024 entries off\n", sa, sb, worst, bad);
    }
    // 2. rate
    float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
    uint4* src; hipMalloc(&src, 4096 * sizeof(uint4));
    unsigned* h = (unsigned*)malloc(4096 * 16);
    for (int i = 0; i < 4096 * 4; ++i) { unsigned r1 = rand(), r2 = rand(); h[i] = ((0x3c00u + (r1 & 0x7ff)) | ((r1 >> 15) & 1) << 15) | (((0x3c00u + (r2 & 0x7ff)) | ((r2 >> 15) & 1) << 15) << 16); }
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        time_rate<0>(src, d, "six bf16 terms (24 x 32x32x16 bf16)");
        time_rate<1>(src, d, "two bf16 + four fp8 terms (8 bf16 + 4 f8f6f4)");
        time_rate<2>(src, d, "six fp8 terms (6 x 32x32x64 f8f6f4)");
    }
    return 0;
}
