// Standalone timing lab for wphase/hphase variants (GPU box only; not part of the product).
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DLAB -o /tmp/lab tools/lab/lab_wphase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#define NMFX_LAB 1
#include "../../nmf_amd/csrc/kernels_products.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float time_it(const char* name, int reps, std::function<void()> f, double flops, double bytes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-34s %8.1f us  %7.1f TF/s  %7.1f GB/s\n", name, ms * 1e3, flops / ms / 1e9, bytes / ms / 1e6);
    return ms;
}

int main(int argc, char** argv) {
    const int64_t m = 16384, n = 8192; constexpr int KP = 64;
    const int ws = argc > 1 ? atoi(argv[1]) : 3, hs = argc > 2 ? atoi(argv[2]) : 4;
    const int64_t pad = argc > 3 ? atoll(argv[3]) : 0, ldv = n + pad;
    printf("pad=%ld floats\n", (long)pad);
    float *V, *W, *H, *A, *B, *G; double* obj; int* flag;
    CK(hipMalloc(&V, m * ldv * 4)); CK(hipMalloc(&W, m * KP * 4)); CK(hipMalloc(&H, KP * n * 4));
    CK(hipMalloc(&A, (size_t)8 * m * KP * 4)); CK(hipMalloc(&B, (size_t)16 * KP * n * 4)); CK(hipMalloc(&G, 16 * KP * KP * 4));
    CK(hipMalloc(&obj, 1 << 16)); CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    std::vector<float> hv(m * ldv); for (auto& x : hv) x = (float)rand() / RAND_MAX;
    CK(hipMemcpy(V, hv.data(), m * ldv * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hv.data(), m * KP * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(H, hv.data() + 12345, KP * n * 4, hipMemcpyHostToDevice));
    const double f2 = 2.0 * m * n * KP, vb = (double)m * n * 4;
    dim3 wg((unsigned)(m / 64), ws), hg((unsigned)(n / 64), hs), blk(256);
    const size_t wshm = (2 * KP * 64 + 4 * 16 * 64) * 4, hshm = KP * 64 * 4;
    time_it("wphase A+obj", 20, [&] { hipLaunchKernelGGL((wphase_kernel<KP, true, true, false>), wg, blk, wshm, 0, V, ldv, W, H, n, A, obj, m, (int)(n / 64), flag, (const int*)nullptr); }, 2 * f2, vb);
    time_it("wphase A only", 20, [&] { hipLaunchKernelGGL((wphase_kernel<KP, true, false, false>), wg, blk, wshm, 0, V, ldv, W, H, n, A, obj, m, (int)(n / 64), flag, (const int*)nullptr); }, f2, vb);
    time_it("wphase obj only", 20, [&] { hipLaunchKernelGGL((wphase_kernel<KP, false, true, false>), wg, blk, wshm, 0, V, ldv, W, H, n, A, obj, m, (int)(n / 64), flag, (const int*)nullptr); }, f2, vb);
    time_it("wphase KL A+obj", 20, [&] { hipLaunchKernelGGL((wphase_kernel<KP, true, true, true>), wg, blk, wshm, 0, V, ldv, W, H, n, A, obj, m, (int)(n / 64), flag, (const int*)nullptr); }, 2 * f2, vb);
    time_it("hphase +G", 20, [&] { hipLaunchKernelGGL((hphase_kernel<KP, true>), hg, blk, hshm, 0, V, ldv, W, B, G, n, m, flag, (const int*)nullptr); }, f2, vb);
    time_it("hphase", 20, [&] { hipLaunchKernelGGL((hphase_kernel<KP, false>), hg, blk, hshm, 0, V, ldv, W, B, G, n, m, flag, (const int*)nullptr); }, f2, vb);
#ifdef LAB_EXTRA
    LAB_EXTRA
#endif
    CK(hipDeviceSynchronize());
    return 0;
}
