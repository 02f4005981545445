// How fast can one MI355X read 512 MiB?  (the ceiling of the H phase: global_load_dwordx4 straight into registers,
// nothing else)   hipcc --offload-arch=gfx950 -O3 tools/lab/stream_read.hip -o /tmp/stream_read && /tmp/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// contiguous per block: block b streams its own chunk (like a product block streams its tiles)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd_chunk(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
    const size_t per = n16 / gridDim.x;
    const u32x4* q = p + per * blockIdx.x;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per; i += UNROLL * 256) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(q + i + u * 256) : q[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename K> void run(const char* name, K kern, int grid, const u32x4* d, size_t n16, unsigned* o) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, n16, o);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, n16, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s grid %5d: %7.1f us  %6.2f TB/s\n", name, grid, ms / reps * 1e3, n16 * 16.0 / (ms / reps * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = 512ull << 20, n16 = bytes / 16;
    u32x4* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(d, 1, bytes);
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        run("strided u8", rd<8, false>, grid, d, n16, o);
        run("strided u8 nt", rd<8, true>, grid, d, n16, o);
        run("chunk u8", rd_chunk<8, false>, grid, d, n16, o);
        run("chunk u8 nt", rd_chunk<8, true>, grid, d, n16, o);
    }
    run("chunk u16 nt", rd_chunk<16, true>, 256, d, n16, o);
    run("chunk u16 nt", rd_chunk<16, true>, 512, d, n16, o);
    return 0;
}
