// Probe: which lanes v_permlane32_swap / v_permlane16_swap (gfx950) move -- the cross-row broadcast of the in-wave pivots.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/permlane_probe.hip -o tools/lab/permlane_probe && tools/lab/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int QSRC> __device__ __forceinline__ unsigned bcast_row_u32(unsigned v) {
    auto h = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    const unsigned z = (QSRC >= 2) ? h[1] : h[0];
    auto g = __builtin_amdgcn_permlane16_swap(z, z, false, false);
    return (QSRC & 1) ? g[1] : g[0];
}
__global__ void k(unsigned* out) {
    const unsigned x = threadIdx.x;
    out[threadIdx.x] = bcast_row_u32<0>(x);
    out[64 + threadIdx.x] = bcast_row_u32<1>(x);
    out[128 + threadIdx.x] = bcast_row_u32<2>(x);
    out[192 + threadIdx.x] = bcast_row_u32<3>(x);
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int q = 0; q < 4; ++q)
        for (int l = 0; l < 64; ++l) bad += h[64 * q + l] != (unsigned)(16 * q + (l & 15));
    for (int q = 0; q < 4; ++q) { printf("row %d:", q); for (int l = 0; l < 64; l += 5) printf(" %u", h[64 * q + l]); printf("\n"); }
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
