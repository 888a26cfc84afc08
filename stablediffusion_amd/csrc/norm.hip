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
        // eight independent 16-byte loads in flight per thread (HBM latency, not VALU, is the limit)
        for (; pix + 7L * rows_par < p1; pix += 8L * rows_par) {
            h8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const h8*>(base + (pix + (long)u * rows_par) * ldx);
#pragma unroll
            for (int u = 0; u < 8; ++u)
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

// Pass 2: y = act((x - mean) * rstd * gamma + beta).  One block = `rows_per` pixels of one sample,
// at most GN_APPLY_NV 16-byte chunks per thread, all loaded BEFORE the prologue that turns the slab
// partials into mean / rstd and the per-channel scale / shift (fixed-order reduction, no separate
// finalize launch), so the prologue's dependent L2 round trips hide under the x loads.
constexpr int GN_APPLY_NV = 12;
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x, long ldx,
                                                       const float* __restrict__ part,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       half_t* __restrict__ y, long ldy, long HW,
                                                       int C, int G, int S, int rows_per, float eps, int silu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);  // [C] scale
    float* sh = sc + C;                          // [C] shift
    float* st = sh + C;                          // [G][2] mean, rstd
    float* red = st + 2 * G;                     // [256 / G][G][2]
    const int n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G;
    const int CC = C >> 3;
    const long p0 = (long)blockIdx.x * rows_per;
    long p1 = p0 + rows_per;
    if (p1 > HW) p1 = HW;
    const int total = (int)(p1 - p0) * CC;           // 16-byte chunks of this block, <= 256 * GN_APPLY_NV
    const half_t* xb = x + ((long)n * HW + p0) * ldx;
    half_t* yb = y + ((long)n * HW + p0) * ldy;
    // chunk i -> (pixel i / CC, chunk i % CC), advanced by 256 per step without dividing
    const int dq = 256 / CC, dr = 256 - dq * CC;
    h8 v[GN_APPLY_NV];
    {
        int pix = tid / CC, cc = tid - pix * CC;
#pragma unroll
        for (int k = 0; k < GN_APPLY_NV; ++k) {
            if (k * 256 < total) {                   // block-uniform
                if (tid + k * 256 < total) v[k] = *reinterpret_cast<const h8*>(xb + (long)pix * ldx + cc * 8);
                pix += dq; cc += dr;
                if (cc >= CC) { cc -= CC; ++pix; }
            }
        }
    }
    // slab partials -> mean / rstd: 256 / G threads per group sum interleaved slabs, then thread g
    // adds those partial sums in order
    const int parts = 256 / G;
    {
        const int g = tid % G, part_i = tid / G;
        if (part_i < parts) {
            float a = 0.f, b = 0.f;
            const float* src = part + ((long)n * S * G + g) * 2;
            for (int k = part_i; k < S; k += parts) { a += src[(long)k * G * 2]; b += src[(long)k * G * 2 + 1]; }
            red[(part_i * G + g) * 2] = a;
            red[(part_i * G + g) * 2 + 1] = b;
        }
    }
    __syncthreads();
    if (tid < G) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < parts; ++k) { a += red[(k * G + tid) * 2]; b += red[(k * G + tid) * 2 + 1]; }
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
    int pix = tid / CC, cc = tid - pix * CC;
#pragma unroll
    for (int k = 0; k < GN_APPLY_NV; ++k) {
        if (k * 256 < total) {
            if (tid + k * 256 < total) {
                const f4 a0 = *reinterpret_cast<const f4*>(sc + cc * 8), a1 = *reinterpret_cast<const f4*>(sc + cc * 8 + 4);
                const f4 b0 = *reinterpret_cast<const f4*>(sh + cc * 8), b1 = *reinterpret_cast<const f4*>(sh + cc * 8 + 4);
                h8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float f = (float)v[k][e] * a0[e] + b0[e];
                    float g = (float)v[k][e + 4] * a1[e] + b1[e];
                    if (silu) { f = silu_f(f); g = silu_f(g); }
                    o[e] = (half_t)f;
                    o[e + 4] = (half_t)g;
                }
                *reinterpret_cast<h8*>(yb + (long)pix * ldy + cc * 8) = o;
            }
            pix += dq; cc += dr;
            if (cc >= CC) { cc -= CC; ++pix; }
        }
    }
}

// Single-pass GroupNorm for the small feature maps (HW <= 1024): one block per (sample, channel
// unit), unit = lcm(channels-per-group, 8) channels, so 16-byte chunks and groups both tile it.
// The block's whole [HW x unit] panel stays in registers between the statistics and the apply:
// x is read once, y written once, one launch instead of two (the 2-launch form is latency-bound
// here: 10-17 us for tensors an HBM pass moves in 2-5 us).  Thread t owns channel chunk t % UC of
// pixels t / UC + k * PL, so its 8 scale/shift pairs are loop invariants.  Variance is centred
// (second reduction over the registers).  Reductions run in a fixed order: bitwise reproducible.
template <int T, int NV>
__global__ __launch_bounds__(T) void gn_fused_kernel(const half_t* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     half_t* __restrict__ y, long ldy, int HW, int C,
                                                     int cpg, int U, float eps, int silu) {
    constexpr int WAVES = T / 64;
    __shared__ float wred[WAVES][8];
    __shared__ float stat[8];
    const int n = blockIdx.y, c0 = blockIdx.x * U, tid = threadIdx.x;
    const int UC = U >> 3, PL = T / UC;
    const int cchunk = tid % UC, plane = tid / UC;
    const bool active = plane < PL;
    const int ch = cchunk * 8;                       // first channel of this thread inside the unit
    const int gA = ch / cpg;                         // local group of channel e is gA or gA + 1 (cpg >= 8)
    const int split = (gA + 1) * cpg - ch;           // channels e < split belong to gA (cpg even: split even)
    const half_t* xb = x + (long)n * HW * ldx + c0 + ch;
    h8 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int p = plane + k * PL;
        if (active && p < HW) v[k] = *reinterpret_cast<const h8*>(xb + (long)p * ldx);
        else v[k] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    // per-thread sums per channel pair: v_dot2_f32_f16 against (1,1) and against itself (zero-filled
    // slots add nothing, so no guards here)
    float s2[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
    const h2 ones = {(half_t)1.f, (half_t)1.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k * PL < HW) {                           // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2 pr = {v[k][2 * j], v[k][2 * j + 1]};
                s2[j] = __builtin_amdgcn_fdot2(pr, ones, s2[j], false);
                q2[j] = __builtin_amdgcn_fdot2(pr, pr, q2[j], false);
            }
        }
    }
    float a = 0.f, b = 0.f, aq = 0.f, bq = 0.f;      // sums / sums of squares for gA and gA + 1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (2 * j < split) { a += s2[j]; aq += q2[j]; } else { b += s2[j]; bq += q2[j]; }
    }
    // block reduction in a fixed order: lanes (xor tree), then waves (serial)
    float r[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float t = (g == gA ? a : 0.f) + (g == gA + 1 ? b : 0.f);
        float u = (g == gA ? aq : 0.f) + (g == gA + 1 ? bq : 0.f);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { t += __shfl_xor(t, off); u += __shfl_xor(u, off); }
        r[g] = t;
        r[4 + g] = u;
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int g = 0; g < 8; ++g) wred[tid >> 6][g] = r[g];
    }
    __syncthreads();
    if (tid < 4) {
        float t = 0.f, u = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { t += wred[w][tid]; u += wred[w][4 + tid]; }
        const float cnt = (float)HW * (float)cpg;
        const float mean = t / cnt;
        float var = u / cnt - mean * mean;
        var = var < 0.f ? 0.f : var;
        stat[tid] = mean;
        stat[4 + tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    if (!active) return;
    const float mA = stat[gA & 3], mB = stat[(gA + 1) & 3], rA = stat[4 + (gA & 3)], rB = stat[4 + ((gA + 1) & 3)];
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float w = (e < split ? rA : rB) * gamma[c0 + ch + e];
        sc[e] = w;
        sh[e] = beta[c0 + ch + e] - (e < split ? mA : mB) * w;
    }
    half_t* yb = y + (long)n * HW * ldy + c0 + ch;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k * PL < HW) {                           // wave-uniform
            const int p = plane + k * PL;
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[k][e] * sc[e] + sh[e];
                if (silu) f = silu_f(f);
                o[e] = (half_t)f;
            }
            if (p < HW) *reinterpret_cast<h8*>(yb + (long)p * ldy) = o;
        }
    }
}

// unit (channels per fused block) or 0 when the fused form does not apply
inline int gn_fused_unit(long HW, int C, int G) {
    const int cpg = C / G;
    if (HW > 1024 || cpg < 8 || (cpg & 1)) return 0;
    int unit = cpg;
    while (unit % 8 != 0) unit += cpg;
    if (unit / cpg > 4 || unit > C || C % unit != 0) return 0;
    const int UC = unit / 8;
    const int T = HW <= 256 ? 256 : 1024;
    const int NVmax = HW <= 64 ? 4 : 16;
    const int PL = T / UC;
    if (PL < 1 || (long)PL * NVmax < HW) return 0;
    return unit;
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
    if (const int U = gn_fused_unit(HW, C, G)) {
        const dim3 grid(C / U, N);
        const int cpg = C / G;
        if (HW <= 64)
            hipLaunchKernelGGL((gn_fused_kernel<256, 4>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, (int)HW, C, cpg, U, eps, silu);
        else if (HW <= 256)
            hipLaunchKernelGGL((gn_fused_kernel<256, 16>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, (int)HW, C, cpg, U, eps, silu);
        else
            hipLaunchKernelGGL((gn_fused_kernel<1024, 16>), grid, dim3(1024), 0, s, x, ldx, gamma, beta, y, ldy, (int)HW, C, cpg, U, eps, silu);
        SD_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int S = gn_slabs(N, HW, C, G);
    const int CB = gn_block_channels(C, G);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(cdiv(C, CB), S, N), dim3(256), 0, s, x, ldx, scratch, HW, C, G, S, CB);
    // apply: rows per block so that a thread holds <= GN_APPLY_NV chunks, and >= ~512 blocks overall
    const int CC = C / 8;
    long rows_per = (long)256 * GN_APPLY_NV / CC;
    if (rows_per < 1) { set_error("groupnorm: C too large for the apply kernel"); return 1; }
    const long want = cdiv(HW * N, 512);
    if (rows_per > want) rows_per = want > 0 ? want : 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)cdiv(HW, rows_per), N), dim3(256),
                       ((size_t)C * 2 + (size_t)G * 2 + 512) * sizeof(float), s,
                       x, ldx, scratch, gamma, beta, y, ldy, HW, C, G, S, (int)rows_per, eps, silu);
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
