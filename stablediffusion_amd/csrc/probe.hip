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

// L2 -> LDS streaming rate of the LDS-DMA path (buffer_load ... lds, 1 KiB per wave instruction) with EVERY CU
// streaming, as the GEMM kernels' operand rings do: `region` bytes per block are walked `passes` times by the block's
// four waves with `depth` pieces outstanding per wave; shared bit 0: all blocks walk the SAME region (the weight operand),
// else block b walks its own (the activation operand); shared bit 1: ordinary loads into registers instead of LDS-DMA.
// Nothing reads the LDS; the pieces overwrite a small ring.
__global__ __launch_bounds__(256) void probe_dma_kernel(const half_t* __restrict__ src, long region, int passes, int depth, int shared,
                                                        float* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const half_t* base = src + ((shared & 1) ? 0 : (long)blockIdx.x * (region / 2));
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(base), 0, (int)region, 0x00020000);
    const int pieces = (int)(region / 1024);                 // 1 KiB pieces of the region
    const int per_wave = pieces / 4;
    int slot = 0;
    if (shared >= 2) {
        // the same walk through the ordinary vector-memory path (16 bytes per lane into registers), eight loads in flight
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ps = 0; ps < passes; ++ps) {
            for (int i = 0; i + 8 <= per_wave; i += 8) {
                f4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    v[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(((long)((i + q) * 4 + wave)) * 1024 + lane * 16), 0, 0));
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += v[q];
            }
        }
        if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) sink[0] = acc[0];
        return;
    }
    for (int ps = 0; ps < passes; ++ps) {
        for (int i = 0; i < per_wave; ++i) {
            const unsigned off = (unsigned)(((long)(i * 4 + wave)) * 1024 + lane * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + (wave * 32 + slot) * 1024), 16, off, 0, 0, 0);
            if (++slot == 32) slot = 0;
            // keep `depth` pieces in flight: wait until at most depth - 1 are outstanding before the next issue
            switch (depth) {
                case 1: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<float*>(smem)[tid] == 1.2345e-30f) sink[0] = 1.f;     // never true
#endif
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

int probe_dma(long region, int passes, int depth, int shared, float* gbs, hipStream_t s) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (region < 4096 || region % 4096 != 0 || region >= (1L << 31) || passes < 1) { set_error("probe_dma: bad arguments"); return 1; }
    const long total = (shared & 1) ? region : region * cus;
    half_t* src = nullptr;
    float* sink = nullptr;
    SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&src), (size_t)total));
    SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 256));
    (void)hipMemsetAsync(src, 1, (size_t)total, s);
    const int lds = 4 * 32 * 1024;
    SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_dma_kernel, dim3(cus), dim3(256), lds, s, src, region, 1, depth, shared, sink);      // warm-up: fills L2 / MALL
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(probe_dma_kernel, dim3(cus), dim3(256), lds, s, src, region, passes, depth, shared, sink);
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(sink);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return 3; }
    *gbs = (float)((double)region * passes * cus / (ms * 1e-3) / 1e9);
    return 0;
}

}  // namespace sd
