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

constexpr int GN_MAX_CB = 512;   // channels per stats block (upper bound)

// Channels per stats block: a multiple of lcm(channels-per-group, 8) so that no group and no
// 16-byte chunk straddles two blocks.
__host__ __device__ inline int gn_block_channels(int C, int G) {
    const int cpg = C / G;
    int unit = cpg;
    while (unit % 8 != 0) unit += cpg;          // lcm(cpg, 8)
    int cb = unit;
    while (cb + unit <= GN_MAX_CB && cb + unit <= C) cb += unit;
    return cb > C ? C : cb;
}

// Pass 1: per-(n, slab, group) sum and sum of squares.  part[((n*S + s)*G + g)*2 + {0,1}]
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x, long ldx,
                                                       float* __restrict__ part, long HW, int C,
                                                       int G, int S, int CB) {
    __shared__ float red[256 * 16];
    __shared__ float chan[GN_MAX_CB * 2];
    const int n = blockIdx.z, s = blockIdx.y, cb = blockIdx.x;
    const int tid = threadIdx.x;
    const int c0 = cb * CB;
    const int cw = (C - c0 < CB) ? C - c0 : CB;      // channels in this block
    const int ccb = cw >> 3;
    const int rows_par = 256 / ccb;
    const int cc_l = tid % ccb, prow = tid / ccb;
    const long rows_per = (HW + S - 1) / S;
    const long p0 = (long)s * rows_per;
    long p1 = p0 + rows_per;
    if (p1 > HW) p1 = HW;
    float sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
    if (prow < rows_par) {
        const half_t* base = x + ((long)n * HW) * ldx + c0 + cc_l * 8;
        long pix = p0 + prow;
        // four independent 16-byte loads in flight per thread (HBM latency, not VALU, is the limit)
        for (; pix + 3L * rows_par < p1; pix += 4L * rows_par) {
            h8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const h8*>(base + (pix + (long)u * rows_par) * ldx);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e]; sm[e] += f; sq[e] += f * f; }
        }
        for (; pix < p1; pix += rows_par) {
            const h8 v = *reinterpret_cast<const h8*>(base + pix * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sm[e] += f; sq[e] += f * f; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = sm[e]; red[tid * 16 + 8 + e] = sq[e]; }
    __syncthreads();
    // fixed-order reduction over the pixel-parallel rows -> per-channel sums in LDS
    if (prow == 0) {
        for (int r = 1; r < rows_par; ++r) {
            const int o = (r * ccb + cc_l) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sm[e] += red[o + e]; sq[e] += red[o + 8 + e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { chan[(cc_l * 8 + e) * 2] = sm[e]; chan[(cc_l * 8 + e) * 2 + 1] = sq[e]; }
    }
    __syncthreads();
    const int cpg = C / G;
    const int ng = cw / cpg;
    if (tid < ng) {
        float a = 0.f, b = 0.f;
        for (int c = 0; c < cpg; ++c) { a += chan[(tid * cpg + c) * 2]; b += chan[(tid * cpg + c) * 2 + 1]; }
        float* dst = part + (((long)n * S + s) * G + c0 / cpg + tid) * 2;
        dst[0] = a; dst[1] = b;
    }
}

// Pass 2: y = act((x - mean) * rstd * gamma + beta); the slab partials are reduced (fixed order)
// in the prologue of every block -- no separate finalize launch.
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x, long ldx,
                                                       const float* __restrict__ part,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       half_t* __restrict__ y, long ldy, long HW,
                                                       int C, int G, int S, int SA, float eps, int silu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);  // [C] scale
    float* sh = sc + C;                          // [C] shift
    float* st = sh + C;                          // [G][2] mean, rstd
    const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    if (tid < G) {
        float a = 0.f, b = 0.f;
        const float* src = part + ((long)n * S * G + tid) * 2;
        for (int k = 0; k < S; ++k) { a += src[(long)k * G * 2]; b += src[(long)k * G * 2 + 1]; }
        const float cnt = (float)HW * (float)cpg;
        const float mean = a / cnt;
        float var = b / cnt - mean * mean;
        var = var < 0.f ? 0.f : var;
        st[tid * 2] = mean;
        st[tid * 2 + 1] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float a = st[g * 2 + 1] * gamma[c];
        sc[c] = a;
        sh[c] = beta[c] - st[g * 2] * a;
    }
    __syncthreads();
    const int CC = C >> 3;
    const long rows_per = (HW + SA - 1) / SA;
    const long p0 = (long)s * rows_per;
    long p1 = p0 + rows_per;
    if (p1 > HW) p1 = HW;
    const long total = (p1 - p0) * CC;
    const half_t* xb = x + ((long)n * HW + p0) * ldx;
    half_t* yb = y + ((long)n * HW + p0) * ldy;
    long i = tid;
    for (; i + 3 * 256 < total; i += 4 * 256) {     // four loads in flight per thread
        h8 v[4];
        long pixs[4];
        int cs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long ii = i + u * 256;
            pixs[u] = ii / CC;
            cs[u] = (int)(ii - pixs[u] * CC) * 8;
            v[u] = *reinterpret_cast<const h8*>(xb + pixs[u] * ldx + cs[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[u][e] * sc[cs[u] + e] + sh[cs[u] + e];
                if (silu) f = silu_f(f);
                o[e] = (half_t)f;
            }
            *reinterpret_cast<h8*>(yb + pixs[u] * ldy + cs[u]) = o;
        }
    }
    for (; i < total; i += 256) {
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

int gn_slabs(int N, long HW, int C, int G) {
    // aim for ~1024 stats blocks, at least 64 pixels per slab, at most 256 slabs
    const int cblocks = cdiv(C, gn_block_channels(C, G));
    long s = 1024 / ((long)N * cblocks);
    if (s < 1) s = 1;
    const long smax = HW / 64 > 0 ? HW / 64 : 1;
    if (s > smax) s = smax;
    if (s > 256) s = 256;
    return (int)s;
}

}  // namespace

long gn_scratch_floats(int N, long HW, int C, int G) {
    return (long)N * gn_slabs(N, HW, C, G) * G * 2;
}

int launch_groupnorm(const half_t* x, long ldx, const float* gamma, const float* beta, half_t* y,
                     long ldy, int N, long HW, int C, int G, float eps, int silu, float* scratch,
                     hipStream_t s) {
    if (C % 8 != 0 || C % G != 0 || G > 256) { set_error("groupnorm: C must be a multiple of 8 and of groups"); return 1; }
    const int S = gn_slabs(N, HW, C, G);
    const int CB = gn_block_channels(C, G);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(cdiv(C, CB), S, N), dim3(256), 0, s, x, ldx, scratch, HW, C, G, S, CB);
    int SA = (int)(2048 / N);
    if (SA < 1) SA = 1;
    const long samax = HW / 16 > 0 ? HW / 16 : 1;
    if (SA > samax) SA = (int)samax;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(SA, N), dim3(256), ((size_t)C * 2 + (size_t)G * 2) * sizeof(float), s,
                       x, ldx, scratch, gamma, beta, y, ldy, HW, C, G, S, SA, eps, silu);
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
