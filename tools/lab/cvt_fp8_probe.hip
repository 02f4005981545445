// r5 probe: semantics of v_cvt_scalef32_pk_fp8_bf16 (gfx950) -- which way the scale goes, where the two bytes land.
//   hipcc -O3 --offload-arch=gfx950 -o cvt_fp8_probe tools/lab/cvt_fp8_probe.hip && ./cvt_fp8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, unsigned* out, float scale) {
    union { unsigned u; bf16x2 v; } s; s.u = in[threadIdx.x];
    s16x2 old = {(short)0x1111, (short)0x2222};
    union { s16x2 v; unsigned u; } a, b;
    a.v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, s.v, scale, false);
    b.v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, s.v, scale, true);
    out[2 * threadIdx.x] = a.u; out[2 * threadIdx.x + 1] = b.u;
}
static float e4m3(unsigned char c) {
    const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    const float v = e ? std::ldexp(1.f + m / 8.f, e - 7) : std::ldexp(m / 8.f, -6);
    return s ? -v : v;
}
static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
int main() {
    const float vals[8][2] = {{1.f, 2.f}, {0.75f, -3.f}, {0.001953125f, 0.00390625f}, {100.f, 448.f}, {0.3f, 0.7f}, {1e-3f, 5e-4f}, {0.0625f, 0.015625f}, {1.0625f, 1.1875f}};
    unsigned h[8], *d, *o, ho[16];
    for (int i = 0; i < 8; ++i) h[i] = bf(vals[i][0]) | ((unsigned)bf(vals[i][1]) << 16);
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (float scale : {1.f, 2.f, 0.001953125f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, scale);
        hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int i = 0; i < 8; ++i) {
            const unsigned a = ho[2 * i], b = ho[2 * i + 1];
            printf("  (%g, %g): word_sel 0 -> %08x [%g %g | %g %g], word_sel 1 -> %08x\n", vals[i][0], vals[i][1], a, e4m3(a & 255), e4m3((a >> 8) & 255),
                   e4m3((a >> 16) & 255), e4m3(a >> 24), b);
        }
    }
    return 0;
}
