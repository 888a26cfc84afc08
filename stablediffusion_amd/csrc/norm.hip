// GroupNorm(+SiLU) and LayerNorm on NHWC fp16 for gfx950.  HBM-bound kernels: 16-byte loads,
// fp32 statistics, deterministic (no float atomics: partial sums are reduced in a fixed order so
// a batch sharded over GPUs reproduces the unsharded result bit for bit).
//
// Replaces aten group_norm / layer_norm + silu issued by diffusers' ResnetBlock2D, Transformer2DModel,
// BasicTransformerBlock and the VAE decoder under the call sites
// /root/reference/pipelines/sd_unified_pipeline.py:475-482 and :523.
#include "kernels.h"

namespace sd {
namespace {

constexpr int GN_CCB = 64;  // channel chunks (of 8) per block in the stats pass

// Pass 1: per-(n, slab, channel) sum and sum of squares.  part[((n*S + s)*C + c)*2 + {0,1}]
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x, long ldx,
                                                       float* __restrict__ part, long HW, int C,
                                                       int S, int ccb) {
    __shared__ float red[256 * 16];
    const int n = blockIdx.z, s = blockIdx.y, cb = blockIdx.x;
    const int CC = C >> 3;
    const int tid = threadIdx.x;
    const int rows_par = 256 / ccb;
    const int cc_l = tid % ccb, prow = tid / ccb;
    const int cc = cb * ccb + cc_l;
    const long rows_per = (HW + S - 1) / S;
    const long p0 = (long)s * rows_per;
    long p1 = p0 + rows_per;
    if (p1 > HW) p1 = HW;
    float sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
    const bool active = prow < rows_par && cc < CC;
    if (active) {
        const half_t* base = x + ((long)n * HW) * ldx + cc * 8;
        for (long pix = p0 + prow; pix < p1; pix += rows_par) {
            const h8 v = *reinterpret_cast<const h8*>(base + pix * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sm[e] += f; sq[e] += f * f; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = sm[e]; red[tid * 16 + 8 + e] = sq[e]; }
    __syncthreads();
    // fixed-order reduction over the pixel-parallel rows
    if (prow == 0 && cc < CC) {
        for (int r = 1; r < rows_par; ++r) {
            const int o = (r * ccb + cc_l) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sm[e] += red[o + e]; sq[e] += red[o + 8 + e]; }
        }
        float* dst = part + (((long)n * S + s) * C + cc * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dst[e * 2] = sm[e]; dst[e * 2 + 1] = sq[e]; }
    }
}

// Pass 2: one wave per (n, group): reduce slabs x channels-of-group -> mean, rstd.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part,
                                                         float* __restrict__ stats, long HW, int C,
                                                         int G, int S, float eps) {
    const int n = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int cpg = C / G;
    const int total = S * cpg;
    float sm = 0.f, sq = 0.f;
    for (int i = lane; i < total; i += 64) {
        const int s = i / cpg, c = g * cpg + (i - s * cpg);
        const float* src = part + (((long)n * S + s) * C + c) * 2;
        sm += src[0]; sq += src[1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sm += __shfl_xor(sm, off);
        sq += __shfl_xor(sq, off);
    }
    if (lane == 0) {
        const float cnt = (float)HW * (float)cpg;
        const float mean = sm / cnt;
        float var = sq / cnt - mean * mean;
        var = var < 0.f ? 0.f : var;
        stats[((long)n * G + g) * 2] = mean;
        stats[((long)n * G + g) * 2 + 1] = rsqrtf(var + eps);
    }
}

// Pass 3: y = act((x - mean) * rstd * gamma + beta)
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x, long ldx,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       half_t* __restrict__ y, long ldy, long HW,
                                                       int C, int G, int S, int silu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);  // [C] scale
    float* sh = sc + C;                          // [C] shift
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float mean = stats[((long)n * G + g) * 2], rstd = stats[((long)n * G + g) * 2 + 1];
        const float a = rstd * gamma[c];
        sc[c] = a;
        sh[c] = beta[c] - mean * a;
    }
    __syncthreads();
    const int CC = C >> 3;
    const long rows_per = (HW + S - 1) / S;
    const long p0 = (long)s * rows_per;
    long p1 = p0 + rows_per;
    if (p1 > HW) p1 = HW;
    const long total = (p1 - p0) * CC;
    const half_t* xb = x + ((long)n * HW + p0) * ldx;
    half_t* yb = y + ((long)n * HW + p0) * ldy;
    for (long i = tid; i < total; i += 256) {
        const long pix = i / CC;
        const int c = (int)(i - pix * CC) * 8;
        const h8 v = *reinterpret_cast<const h8*>(xb + pix * ldx + c);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e] * sc[c + e] + sh[c + e];
            if (silu) f = silu_f(f);
            o[e] = (half_t)f;
        }
        *reinterpret_cast<h8*>(yb + pix * ldy + c) = o;
    }
}

// LayerNorm: one wave per row, row held in registers (C <= 64 * 8 * LN_MAX chunks).
constexpr int LN_MAX = 4;  // up to 2048 channels
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        half_t* __restrict__ y, long ldy, long rows,
                                                        int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CC = C >> 3;
    h8 v[LN_MAX];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i) {
        const int cc = lane + 64 * i;
        if (cc < CC) {
            v[i] = *reinterpret_cast<const h8*>(x + row * ldx + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += (float)v[i][e];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sm += __shfl_xor(sm, off);
    const float mean = sm / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i) {
        const int cc = lane + 64 * i;
        if (cc < CC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i) {
        const int cc = lane + 64 * i;
        if (cc < CC) {
            const f4 g0 = *reinterpret_cast<const f4*>(gamma + cc * 8);
            const f4 g1 = *reinterpret_cast<const f4*>(gamma + cc * 8 + 4);
            const f4 b0 = *reinterpret_cast<const f4*>(beta + cc * 8);
            const f4 b1 = *reinterpret_cast<const f4*>(beta + cc * 8 + 4);
            h8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)(((float)v[i][e] - mean) * rstd * g0[e] + b0[e]);
                o[e + 4] = (half_t)(((float)v[i][e + 4] - mean) * rstd * g1[e] + b1[e]);
            }
            *reinterpret_cast<h8*>(y + row * ldy + cc * 8) = o;
        }
    }
}

int gn_slabs(int N, long HW, int C) {
    // aim for ~1024 blocks, at least 64 pixels per slab
    const int cblocks = cdiv(C >> 3, GN_CCB);
    long s = 1024 / ((long)N * cblocks);
    if (s < 1) s = 1;
    const long smax = HW / 64 > 0 ? HW / 64 : 1;
    if (s > smax) s = smax;
    return (int)s;
}

}  // namespace

long gn_scratch_floats(int N, long HW, int C, int G) {
    const int S = gn_slabs(N, HW, C);
    return (long)N * S * C * 2 + (long)N * G * 2;
}

int launch_groupnorm(const half_t* x, long ldx, const float* gamma, const float* beta, half_t* y,
                     long ldy, int N, long HW, int C, int G, float eps, int silu, float* scratch,
                     hipStream_t s) {
    if (C % 8 != 0 || C % G != 0) { set_error("groupnorm: C must be a multiple of 8 and of groups"); return 1; }
    const int S = gn_slabs(N, HW, C);
    const int CC = C >> 3;
    const int ccb = CC < GN_CCB ? CC : GN_CCB;
    float* part = scratch;
    float* stats = scratch + (long)N * S * C * 2;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(cdiv(CC, ccb), S, N), dim3(256), 0, s, x, ldx, part, HW, C, S, ccb);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(64), 0, s, part, stats, HW, C, G, S, eps);
    // apply pass: finer slabs for parallelism
    int SA = (int)(2048 / N);
    if (SA < 1) SA = 1;
    const long samax = HW / 16 > 0 ? HW / 16 : 1;
    if (SA > samax) SA = (int)samax;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(SA, N), dim3(256), (size_t)C * 2 * sizeof(float), s, x, ldx,
                       stats, gamma, beta, y, ldy, HW, C, G, SA, silu);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_layernorm(const half_t* x, long ldx, const float* gamma, const float* beta, half_t* y,
                     long ldy, long rows, int C, float eps, hipStream_t s) {
    if (C % 8 != 0 || C > 64 * 8 * LN_MAX) { set_error("layernorm: unsupported C"); return 1; }
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy,
                       rows, C, eps);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sd
