// Shared device/host helpers for the gfx950 (CDNA4, wave64) denoise engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define SD_WAVE 64

namespace sd {

// Thread-local error string behind sd_last_error().
void set_error(const std::string& msg);

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: one process may drive engines on several
// GPUs (HipUNet2DConditionModel(device="cuda:1")), so "already set" is remembered per (launcher, device).
// Setting it twice is harmless, so a racy first call from two threads only costs a repeat.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        return (done.fetch_or(bit) & bit) == 0;
    }
};

}  // namespace sd

#define SD_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            sd::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));           \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

#ifdef __HIPCC__
// x * sigmoid(x) with raw v_exp_f32 / v_rcp_f32 (1 ulp-level error, far below fp16 rounding).
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// Exact-erf GELU (diffusers GEGLU uses F.gelu's default, not the tanh form).  erf through
// Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below fp16 output rounding) with raw
// v_rcp_f32 / v_exp_f32: ~14 VALU ops instead of libm erff's branchy polynomial.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erf_abs = __builtin_fmaf(-poly * t, e, 1.0f);       // erf(|x| / sqrt 2)
    const float erf_signed = __builtin_copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf_signed);
}
// CLIP's quick_gelu: x * sigmoid(1.702 x)
__device__ __forceinline__ float quick_gelu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
// XCD-aware, bijective block-id remap (8 XCDs, round-robin dispatch): blocks that end up on one
// XCD get a contiguous range of tile ids so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}
#endif
