// Box probe (bench.py `box_probe`): two fixed, model-independent microbenchmarks -- a back-to-back
// v_mfma_f32_16x16x32_f16 loop on register operands and a 16-byte-per-lane device copy -- so that a bench
// value can be read against the box (clock / HBM) it ran on.  Boxes of the pool differ by up to ~10 % on one
// binary (MI355X_MICROARCH.md, DVFS give-back item 5); the probe runs in-process, outside the timed region.
#include "kernels.h"

namespace sd {
namespace {

// One wave per SIMD, 16 independent accumulators, operands random in [-1, 1) (zero operands clock higher).
__global__ __launch_bounds__(256) void probe_mfma_kernel(float* sink, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() {
        h = h * 1664525u + 1013904223u;
        return (half_t)(((float)(h >> 8) * (1.0f / 8388608.0f)) - 1.0f);
    };
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[i][e] = rnd(); b[i][e] = rnd(); }
    f4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    f4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 1.2345e-30f) sink[0] = s[0];     // keeps the loop alive, never true
#endif
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

}  // namespace

int probe_mfma(int iters, float* tflops, hipStream_t s) {
    float* sink = nullptr;
    SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 256));
    hipEvent_t e0, e1;
    SD_HIP_CHECK(hipEventCreate(&e0));
    SD_HIP_CHECK(hipEventCreate(&e1));
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(cus), dim3(256), 0, s, sink, iters / 8 + 1);      // warm-up (clock ramp)
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(cus), dim3(256), 0, s, sink, iters);
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return 3; }
    const double flop = (double)cus * 4.0 * (double)iters * 16.0 * (2.0 * 16 * 16 * 32);
    *tflops = (float)(flop / (ms * 1e-3) / 1e12);
    return 0;
}

int probe_copy(long bytes, int iters, float* gbs, hipStream_t s) {
    f4 *src = nullptr, *dst = nullptr;
    const long n = bytes / 16;
    SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&src), (size_t)n * 16));
    if (hipMalloc(reinterpret_cast<void**>(&dst), (size_t)n * 16) != hipSuccess) { (void)hipFree(src); set_error("probe_copy: hipMalloc"); return 3; }
    (void)hipMemsetAsync(src, 1, (size_t)n * 16, s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_copy_kernel, dim3(8192), dim3(256), 0, s, src, dst, n);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe_copy_kernel, dim3(8192), dim3(256), 0, s, src, dst, n);
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(dst);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return 3; }
    *gbs = (float)(2.0 * (double)n * 16.0 * iters / (ms * 1e-3) / 1e9);
    return 0;
}

}  // namespace sd
