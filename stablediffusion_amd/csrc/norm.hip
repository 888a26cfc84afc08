// GroupNorm(+SiLU) and LayerNorm on NHWC fp16 for gfx950.  HBM-bound kernels: 16-byte loads,
// fp32 statistics, deterministic (no float atomics: partial sums are reduced in a fixed order so
// a batch sharded over GPUs reproduces the unsharded result bit for bit).
//
// Replaces aten group_norm / layer_norm + silu issued by diffusers' ResnetBlock2D, Transformer2DModel,
// BasicTransformerBlock and the VAE decoder under the call sites
// /root/reference/pipelines/sd_unified_pipeline.py:475-482 and :523.
#include <cstdlib>

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

// Merge of two (count, mean, M2) summaries (Chan et al.): exact up to rounding whatever |mean| / std is,
// unlike E[x^2] - mean^2 (a VAE group spans up to 4 M elements with eps = 1e-6).
__device__ __forceinline__ void chan_merge(float& nA, float& mA, float& qA, float nB, float mB, float qB) {
    if (nB <= 0.f) return;
    const float n = nA + nB;
    const float d = mB - mA;
    const float f = nB / n;
    mA += d * f;
    qA += qB + d * d * nA * f;
    nA = n;
}

// Pass 1: per-(n, slab, group) statistics as (mean, M2 = sum (x - mean)^2) of the slab's
// rows_per x channels-per-group elements:  part[((n*S + s)*G + g)*2 + {0,1}].  Each thread accumulates
// its pixels per channel SHIFTED by the first value it sees for that channel (sums of x - x0 and their
// squares cancel nothing even when |mean| >> std); thread, channel and slab summaries are then merged
// with chan_merge in a fixed order.
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x, long ldx,
                                                       float* __restrict__ part, long HW, int C,
                                                       int G, int S, int CB) {
    __shared__ float red[256 * 16];
    __shared__ float rcnt[256];
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
    float sm[8], sq[8], piv[8];
    float cnt = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; piv[e] = 0.f; }
    if (prow < rows_par && p0 + prow < p1) {
        const half_t* base = x + ((long)n * HW) * ldx + c0 + cc_l * 8;
        long pix = p0 + prow;
        {
            const h8 v0 = *reinterpret_cast<const h8*>(base + pix * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) piv[e] = (float)v0[e];
        }
        // eight independent 16-byte loads in flight per thread (HBM latency, not VALU, is the limit)
        for (; pix + 7L * rows_par < p1; pix += 8L * rows_par) {
            h8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const h8*>(base + (pix + (long)u * rows_par) * ldx);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e] - piv[e]; sm[e] += f; sq[e] += f * f; }
            cnt += 8.f;
        }
        for (; pix < p1; pix += rows_par) {
            const h8 v = *reinterpret_cast<const h8*>(base + pix * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e] - piv[e]; sm[e] += f; sq[e] += f * f; }
            cnt += 1.f;
        }
    }
    // thread summary per channel: mean = pivot + S1 / n, M2 = S2 - S1^2 / n
    const float icnt = cnt > 0.f ? 1.0f / cnt : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float d = sm[e] * icnt;
        red[tid * 16 + e] = piv[e] + d;
        red[tid * 16 + 8 + e] = sq[e] - sm[e] * d;
    }
    rcnt[tid] = cnt;
    __syncthreads();
    // fixed-order merge over the pixel-parallel rows -> per-channel (mean, M2) in LDS
    float ctot = 0.f;
    if (prow == 0) {
        float mA[8], qA[8];
        float nA = cnt;
#pragma unroll
        for (int e = 0; e < 8; ++e) { mA[e] = red[tid * 16 + e]; qA[e] = red[tid * 16 + 8 + e]; }
        for (int r = 1; r < rows_par; ++r) {
            const int t = r * ccb + cc_l;
            const float nB = rcnt[t];
            float nn = nA;
#pragma unroll
            for (int e = 0; e < 8; ++e) { nn = nA; chan_merge(nn, mA[e], qA[e], nB, red[t * 16 + e], red[t * 16 + 8 + e]); }
            nA = nn;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { chan[(cc_l * 8 + e) * 2] = mA[e]; chan[(cc_l * 8 + e) * 2 + 1] = qA[e]; }
        ctot = nA;
    }
    __syncthreads();
    const int cpg = C / G;
    const int ng = cw / cpg;
    if (tid < ng) {
        const float per = (float)(p1 - p0);          // every channel of the slab saw p1 - p0 pixels
        float nA = per, mA = chan[(tid * cpg) * 2], qA = chan[(tid * cpg) * 2 + 1];
        for (int c = 1; c < cpg; ++c) chan_merge(nA, mA, qA, per, chan[(tid * cpg + c) * 2], chan[(tid * cpg + c) * 2 + 1]);
        float* dst = part + (((long)n * S + s) * G + c0 / cpg + tid) * 2;
        dst[0] = mA; dst[1] = qA;
    }
    (void)ctot;
}

// Merges the S slab summaries of every (sample, group) into one (S' = 1 form of `part`): used ahead of the
// apply pass when a producing convolution wrote one summary per 256-pixel tile (up to 1024 per image at
// 512 x 512), which would be too many for every apply block to merge in its prologue.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int S,
                                                          int G, long rows_per, long HW, int cpg) {
    __shared__ float rn[256], rm[256], rq[256];
    const int n = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    float nA = 0.f, mA = 0.f, qA = 0.f;
    for (int k = tid; k < S; k += 256) {
        long rows = HW - (long)k * rows_per;
        if (rows > rows_per) rows = rows_per;
        const float2 v = *reinterpret_cast<const float2*>(part + (((long)n * S + k) * G + g) * 2);
        chan_merge(nA, mA, qA, (float)rows * (float)cpg, v.x, v.y);
    }
    rn[tid] = nA; rm[tid] = mA; rq[tid] = qA;
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 256 && k < S; ++k) chan_merge(nA, mA, qA, rn[k], rm[k], rq[k]);
        float* dst = out + ((long)n * G + g) * 2;
        dst[0] = mA; dst[1] = qA;
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
                                                       int C, int G, int S, long stat_rows, int rows_per, float eps,
                                                       int silu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);  // [C] scale
    float* sh = sc + C;                          // [C] shift
    float* st = sh + C;                          // [G][2] mean, rstd
    float* red = st + 2 * G;                     // [256 / G][G][3]
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
    // slab summaries (mean, M2) -> mean / rstd: 256 / G threads per group merge interleaved slabs, then
    // thread g merges those in order (chan_merge: fixed order, no cancellation)
    const int parts = 256 / G;
    {
        const int g = tid % G, part_i = tid / G;
        if (part_i < parts) {
            float nA = 0.f, mA = 0.f, qA = 0.f;
            const float* src = part + ((long)n * S * G + g) * 2;
            for (int k = part_i; k < S; k += parts) {
                long rows = HW - (long)k * stat_rows;
                if (rows > stat_rows) rows = stat_rows;
                chan_merge(nA, mA, qA, (float)rows * (float)cpg, src[(long)k * G * 2], src[(long)k * G * 2 + 1]);
            }
            red[(part_i * G + g) * 3] = nA;
            red[(part_i * G + g) * 3 + 1] = mA;
            red[(part_i * G + g) * 3 + 2] = qA;
        }
    }
    __syncthreads();
    if (tid < G) {
        float nA = red[tid * 3], mA = red[tid * 3 + 1], qA = red[tid * 3 + 2];
        for (int k = 1; k < parts; ++k) chan_merge(nA, mA, qA, red[(k * G + tid) * 3], red[(k * G + tid) * 3 + 1], red[(k * G + tid) * 3 + 2]);
        const float var = qA / ((float)HW * (float)cpg);
        st[tid * 2] = mA;
        st[tid * 2 + 1] = rsqrtf((var < 0.f ? 0.f : var) + eps);
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


// Apply pass, channel-blocked (round 3): one block = NV x rows_par pixels x one channel block of CB channels (CB as
// in the statistics kernel: a multiple of lcm(channels per group, 8), at most 512).  Against gn_apply_kernel, whose
// every block built scale / shift for ALL C channels and merged the summaries of ALL groups in its prologue, a block's
// prologue here only covers its own groups, and a thread keeps the scale / shift of its eight channels in registers
// (no LDS table): the small maps (8 x 8 ... 32 x 32, where the prologue was most of the block) run at a few us.
template <int NV>
__global__ __launch_bounds__(256) void gn_apply2_kernel(const half_t* __restrict__ x, long ldx, const float* __restrict__ part,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        half_t* __restrict__ y, long ldy, long HW, int C, int G, int S,
                                                        long stat_rows, int CB, float eps, int silu) {
    __shared__ float st[128 * 2];
    __shared__ float red[256 * 3];
    const int n = blockIdx.z, tid = threadIdx.x;
    const int c0 = blockIdx.y * CB;
    const int cw = (C - c0 < CB) ? C - c0 : CB;
    const int CCB = cw >> 3;
    const int rows_par = 256 / CCB;
    const int chunk = tid % CCB, prow = tid / CCB;
    const bool active = prow < rows_par;
    const long row0 = (long)blockIdx.x * rows_par * NV;
    const half_t* xb = x + ((long)n * HW) * ldx + c0 + chunk * 8;
    const int cpg = C / G;
    const int ng = cw / cpg, g0 = c0 / cpg;
    // The prologue runs in EVERY block (1368 of them on 32768 x 320) and all blocks of an image read the SAME 4 KB of
    // summaries at the same time: with all four waves loading them (pivot + 2 per thread) the requests pile up on a few
    // L2 lines, and with S = 0 (nothing to load) the kernel ran 10.5 instead of 14.7 us (SD_GN_DBG_S0 experiment,
    // profiles/r03_groupnorm_apply.txt).  They are fetched by as few waves as the count allows (below) and merged without
    // divisions:
    // equal-count summaries (every slab has stat_rows rows), p = the first slab's mean,
    //   a = sum (m_k - p),  b = sum (m_k - p)^2,  q = sum M2_k:   mean = p + a / S,   M2 = q + cnt (b - a^2 / S)
    // (the shift by p keeps b - a^2 / S free of cancellation whatever |mean| / std is).  A ragged last slab takes the
    // general chan_merge path.
    const bool equal = HW % stat_rows == 0;
    constexpr int KPRE = 4;
    // few summaries (S * ng <= 128: the 32 x 32 maps): wave 0 alone fetches and merges them -- no second barrier, the
    // other waves go straight to the one below (10.3 -> 7.7 us on 8192 x 640); many (64 x 64 maps, S = 16): all four
    // waves share the fetch, two per thread (one wave with eight per lane: 17.2 us against 12.9)
    const int LW = (equal && S * ng <= 128 && ng <= 64) ? 1 : 4;
    const int parts1 = 64 * LW / ng;                         // threads per group (ng <= 128: launcher)
    const int gi = tid % ng, pi = tid / ng;
    const bool fetch = tid < 64 * LW;                        // wave-uniform
    const bool loader = fetch && pi < parts1;
    const float* src = part + ((long)n * S * G + g0 + gi) * 2;
    // (every load below is unconditional on a clamped index: a load under a lane condition becomes a branch, and the
    //  compiler put s_waitcnt vmcnt(0) between such branches -- the loads went out one round trip at a time)
    float2 pre[KPRE];
    float pivot = 0.f;
    if (fetch) {
        pivot = src[0];
#pragma unroll
        for (int j = 0; j < KPRE; ++j) {
            int k = pi + j * parts1;
            k = k < S ? k : (S > 0 ? S - 1 : 0);
            pre[j] = *reinterpret_cast<const float2*>(src + (long)k * G * 2);
        }
    }
    const int cch = active ? c0 + chunk * 8 : 0;
    const f4 ga = *reinterpret_cast<const f4*>(gamma + cch), gb = *reinterpret_cast<const f4*>(gamma + cch + 4);
    const f4 ba = *reinterpret_cast<const f4*>(beta + cch), bb = *reinterpret_cast<const f4*>(beta + cch + 4);
    h8 v[NV];
    const half_t* xt = active ? xb : x + ((long)n * HW) * ldx;        // idle threads (256 % chunks-per-row of them) read a valid address
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        long pr = row0 + prow + (long)k * rows_par;
        pr = pr < HW ? pr : HW - 1;
        v[k] = *reinterpret_cast<const h8*>(xt + (active ? pr : 0) * ldx);
    }
    if (equal) {
        float sa = 0.f, sb = 0.f, sq = 0.f;
        if (fetch) {
            if (loader) {
#pragma unroll
                for (int j = 0; j < KPRE; ++j)
                    if (pi + j * parts1 < S) { const float d = pre[j].x - pivot; sa += d; sb += d * d; sq += pre[j].y; }
                for (int k = pi + KPRE * parts1; k < S; k += parts1) {      // (more than KPRE per thread: the VAE's big maps)
                    const float d = src[(long)k * G * 2] - pivot;
                    sa += d; sb += d * d; sq += src[(long)k * G * 2 + 1];
                }
            }
            red[tid * 3] = sa; red[tid * 3 + 1] = sb; red[tid * 3 + 2] = sq;
        }
        if (LW == 4) __syncthreads();                        // (LW == 1: writer and reader are the same wave, LDS is in order)
        if (tid < ng) {
            sa = 0.f; sb = 0.f; sq = 0.f;
            for (int k = 0; k < parts1; ++k) { sa += red[(k * ng + tid) * 3]; sb += red[(k * ng + tid) * 3 + 1]; sq += red[(k * ng + tid) * 3 + 2]; }
            const float invS = __builtin_amdgcn_rcpf((float)S);
            const float dm = sa * invS;
            const float cnt = (float)stat_rows * (float)cpg;
            const float m2 = sq + cnt * (sb - sa * dm);
            const float var = m2 * __builtin_amdgcn_rcpf((float)HW * (float)cpg);
            st[tid * 2] = pivot + dm;
            st[tid * 2 + 1] = rsqrtf((var < 0.f ? 0.f : var) + eps);
        }
    } else {
        const int parts = 256 / ng;
        if (pi < parts) {
            float nA = 0.f, mA = 0.f, qA = 0.f;
            for (int k = pi; k < S; k += parts) {
                long rows = HW - (long)k * stat_rows;
                if (rows > stat_rows) rows = stat_rows;
                chan_merge(nA, mA, qA, (float)rows * (float)cpg, src[(long)k * G * 2], src[(long)k * G * 2 + 1]);
            }
            red[(pi * ng + gi) * 3] = nA; red[(pi * ng + gi) * 3 + 1] = mA; red[(pi * ng + gi) * 3 + 2] = qA;
        }
        __syncthreads();
        if (tid < ng) {
            float nA = red[tid * 3], mA = red[tid * 3 + 1], qA = red[tid * 3 + 2];
            for (int k = 1; k < parts; ++k) chan_merge(nA, mA, qA, red[(k * ng + tid) * 3], red[(k * ng + tid) * 3 + 1], red[(k * ng + tid) * 3 + 2]);
            const float var = qA / ((float)HW * (float)cpg);
            st[tid * 2] = mA;
            st[tid * 2 + 1] = rsqrtf((var < 0.f ? 0.f : var) + eps);
        }
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
    if (cpg >= 8 || cpg == 4) {
        // the thread's eight channels lie in at most two groups
        const int cl = chunk * 8;
        const int gA = (int)(((float)cl + 0.5f) * __builtin_amdgcn_rcpf((float)cpg));
        const int split = (gA + 1) * cpg - cl;                 // channels e < split are in group gA
        const int gB = split < 8 ? gA + 1 : gA;
        const float mA_ = st[gA * 2], rA = st[gA * 2 + 1], mB_ = st[gB * 2], rB = st[gB * 2 + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = (e < split ? rA : rB) * (e < 4 ? ga[e] : gb[e - 4]);
            sc[e] = w;
            sh[e] = (e < 4 ? ba[e] : bb[e - 4]) - (e < split ? mA_ : mB_) * w;
        }
    } else {
        const float inv_cpg = 1.0f / (float)cpg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int gl = (int)(((float)(chunk * 8 + e) + 0.5f) * inv_cpg);
            const float w = st[gl * 2 + 1] * (e < 4 ? ga[e] : gb[e - 4]);
            sc[e] = w;
            sh[e] = (e < 4 ? ba[e] : bb[e - 4]) - st[gl * 2] * w;
        }
    }
    half_t* yb = y + ((long)n * HW) * ldy + c0 + chunk * 8;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const long pr = row0 + prow + (long)k * rows_par;
        if (pr < HW) {
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[k][e] * sc[e] + sh[e];
                if (silu) f = silu_f(f);
                o[e] = (half_t)f;
            }
            *reinterpret_cast<h8*>(yb + pr * ldy) = o;
        }
    }
}

// Statistics pass with one PIVOT PER GROUP (round 3).  gn_stats_kernel keeps a pivot per (thread, channel) and then
// merges thread, channel and slab summaries with Chan's formula -- exact, but the merges (divisions, done by the
// 256 / rows_par threads of pixel row 0 in a serial loop) took longer than the pass over x.  Here every element of a
// group is shifted by the SAME value, the group's first channel at the slab's first pixel (any sample of the group is
// within a few standard deviations of its mean, which is all the shift has to achieve), so thread partials are plain
// sums: added per group in a fixed order, then one (mean, M2) conversion.  Needs every 16-byte chunk inside at most
// two groups (channels per group >= 8, or == 4).
__global__ __launch_bounds__(256) void gn_stats2_kernel(const half_t* __restrict__ x, long ldx, float* __restrict__ part, long HW,
                                                        int C, int G, int S, int CB) {
    __shared__ float red[256 * 4];
    const int n = blockIdx.z, s = blockIdx.y, cb = blockIdx.x, tid = threadIdx.x;
    const int c0 = cb * CB;
    const int cw = (C - c0 < CB) ? C - c0 : CB;
    const int CCB = cw >> 3;
    const int rows_par = 256 / CCB;
    const int chunk = tid % CCB, prow = tid / CCB;
    const long rows_per = (HW + S - 1) / S;
    const long p0 = (long)s * rows_per;
    long p1 = p0 + rows_per;
    if (p1 > HW) p1 = HW;
    const int cpg = C / G;
    const int ch = chunk * 8;                                // first channel of the chunk inside the block
    const int gA = ch / cpg;                                 // local group of the chunk's first channel
    const int split = (gA + 1) * cpg - ch;                   // channels e < split belong to gA, the rest to gA + 1
    const half_t* img = x + ((long)n * HW) * ldx + c0;
    const bool twog = split < 8;
    const float pivA = (float)img[p0 * ldx + gA * cpg];
    const float pivB = twog ? (float)img[p0 * ldx + (gA + 1) * cpg] : 0.f;
    float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
    if (prow < rows_par) {
        const half_t* base = img + ch;
        long pix = p0 + prow;
        for (; pix + 7L * rows_par < p1; pix += 8L * rows_par) {
            h8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const h8*>(base + (pix + (long)u * rows_par) * ldx);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e < split) { const float f = (float)v[u][e] - pivA; sa += f; qa += f * f; }
                    else { const float f = (float)v[u][e] - pivB; sb += f; qb += f * f; }
                }
        }
        for (; pix < p1; pix += rows_par) {
            const h8 v = *reinterpret_cast<const h8*>(base + pix * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < split) { const float f = (float)v[e] - pivA; sa += f; qa += f * f; }
                else { const float f = (float)v[e] - pivB; sb += f; qb += f * f; }
            }
        }
    }
    *reinterpret_cast<f4*>(red + tid * 4) = f4{sa, qa, sb, qb};
    __syncthreads();
    const int ng = cw / cpg;
    if (tid < ng) {
        const int cfirst = (tid * cpg) / 8, clast = ((tid + 1) * cpg - 1) / 8;
        float sm = 0.f, sq = 0.f;
        for (int c8 = cfirst; c8 <= clast; ++c8) {
            const int off = ((c8 * 8) / cpg == tid) ? 0 : 2;             // this group is the chunk's first, or its second
            for (int r = 0; r < rows_par; ++r) {
                const float2 v = *reinterpret_cast<const float2*>(red + (r * CCB + c8) * 4 + off);
                sm += v.x; sq += v.y;
            }
        }
        const float cnt = (float)(p1 - p0) * (float)cpg;
        const float d = sm / cnt;
        const float piv = (float)img[p0 * ldx + tid * cpg];
        float* dst = part + (((long)n * S + s) * G + c0 / cpg + tid) * 2;
        dst[0] = piv + d;
        dst[1] = sq - sm * d;
    }
}

// Single-pass GroupNorm for the small feature maps (HW <= 1024): one block per (sample, channel
// unit), unit = lcm(channels-per-group, 8) channels, so 16-byte chunks and groups both tile it.
// The block's whole [HW x unit] panel stays in registers between the statistics and the apply:
// x is read once, y written once, one launch instead of two (the 2-launch form is latency-bound
// here: 10-17 us for tensors an HBM pass moves in 2-5 us).  Thread t owns channel chunk t % UC of
// pixels t / UC + k * PL, so its 8 scale/shift pairs are loop invariants.  Variance is centred: a
// second reduction over the registers, sum (x - mean)^2.  Reductions run in a fixed order: bitwise
// reproducible.
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
    // Two passes over the register-resident panel: group means first, then sum (x - mean)^2 -- exact
    // whatever |mean| / std is (the E[x^2] - mean^2 form of round 1 lost the variance to cancellation
    // for |mean| >> std).  Per-thread sums per channel pair with v_dot2_f32_f16 against (1,1);
    // zero-filled slots add nothing to the sums and are masked out of the squares.
    float s2[4] = {0.f, 0.f, 0.f, 0.f};
    const h2 ones = {(half_t)1.f, (half_t)1.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k * PL < HW) {                           // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2 pr = {v[k][2 * j], v[k][2 * j + 1]};
                s2[j] = __builtin_amdgcn_fdot2(pr, ones, s2[j], false);
            }
        }
    }
    // block reduction in a fixed order: lanes (xor tree), then waves (serial)
    auto block_reduce4 = [&](float a, float b, float* dst4) {
        float r[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t = (g == gA ? a : 0.f) + (g == gA + 1 ? b : 0.f);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            r[g] = t;
        }
        __syncthreads();                             // wred free again (second use)
        if ((tid & 63) == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) wred[tid >> 6][g] = r[g];
        }
        __syncthreads();
        if (tid < 4) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += wred[w][tid];
            dst4[tid] = t;
        }
        __syncthreads();
    };
    float a = 0.f, b = 0.f;                          // sums for gA and gA + 1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (2 * j < split) a += s2[j]; else b += s2[j];
    }
    const float cnt = (float)HW * (float)cpg;
    block_reduce4(a, b, stat);
    const float mA = stat[gA & 3] / cnt, mB = stat[(gA + 1) & 3] / cnt;
    float qa = 0.f, qb = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k * PL < HW && active && plane + k * PL < HW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dlt = (float)v[k][e] - (e < split ? mA : mB);
                if (e < split) qa += dlt * dlt; else qb += dlt * dlt;
            }
        }
    }
    block_reduce4(qa, qb, stat + 4);
    if (!active) return;
    const float rA = rsqrtf(stat[4 + (gA & 3)] / cnt + eps), rB = rsqrtf(stat[4 + ((gA + 1) & 3)] / cnt + eps);
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
    // 32 x 32 maps: the statistics + apply pair is never slower than the single kernel since round 3's apply kernel
    // (C = 960: 15.9 vs 25.6 us, 1920: 23.5 vs 28.8, 640: 15.2 vs 15.4), so the single-kernel form is for HW <= 512
    static const long hw_max = getenv("SD_GN_FUSED_MAX_HW") ? atol(getenv("SD_GN_FUSED_MAX_HW")) : 512;
    if (HW > hw_max || cpg < 8 || (cpg & 1)) return 0;
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

// Per-row sum and sum of squares (one wave per row): the stand-alone producer of the LayerNorm
// statistics for tensors whose GEMM could not emit them from its epilogue (split-K launches).
__global__ __launch_bounds__(256) void row_stats_kernel(const half_t* __restrict__ x, long ldx, float* __restrict__ stat,
                                                        long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CC = C >> 3;
    float sm = 0.f, sq = 0.f;
    for (int cc = lane; cc < CC; cc += 64) {
        const h8 v = *reinterpret_cast<const h8*>(x + row * ldx + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sm += f; sq += f * f; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sm += __shfl_xor(sm, off); sq += __shfl_xor(sq, off); }
    if (lane == 0) { stat[row * 2] = sm; stat[row * 2 + 1] = sq; }
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

// Summaries from the producer pay from 32 x 32 maps up: below, a GroupNorm launch costs its ~5 us floor whatever it
// reads, the single-kernel form (x read once, statistics in registers) is already there, and the producer's extra
// work (a split-K reduction kernel that also reduces per group: +5 us) is a net loss (rocprofv3, round 3).
bool gn_wants_stats(long HW, int C, int G) { return gn_fused_unit(HW, C, G) == 0 || HW >= 1024; }

static int launch_stats_pass(const half_t* x, long ldx, int N, long HW, int C, int G, int S, float* scratch, hipStream_t s) {
    const int CB = gn_block_channels(C, G);
    const int cpg = C / G;
    static const bool old = getenv("SD_GN_OLD") != nullptr;         // A/B switch: round-2 kernels
    if (!old && (cpg >= 8 || cpg == 4))
        hipLaunchKernelGGL(gn_stats2_kernel, dim3(cdiv(C, CB), S, N), dim3(256), 0, s, x, ldx, scratch, HW, C, G, S, CB);
    else
        hipLaunchKernelGGL(gn_stats_kernel, dim3(cdiv(C, CB), S, N), dim3(256), 0, s, x, ldx, scratch, HW, C, G, S, CB);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_groupnorm(const half_t* x, long ldx, const float* gamma, const float* beta, half_t* y,
                     long ldy, int N, long HW, int C, int G, float eps, int silu, float* scratch,
                     hipStream_t s, const GnStats* pre) {
    if (C % 8 != 0 || C % G != 0 || G > 256) { set_error("groupnorm: C must be a multiple of 8 and of groups"); return 1; }
    static const bool old = getenv("SD_GN_OLD") != nullptr;             // A/B switch: round-2 kernels and choices
    const bool have_pre = pre && pre->part && (old ? gn_fused_unit(HW, C, G) == 0 : gn_wants_stats(HW, C, G));
    // summaries already there (left by the producing convolution's epilogue): apply pass only, at every map size;
    // otherwise the small maps take the single-kernel form (x read once), the big ones statistics + apply
    if (!have_pre) {
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
    }
    int S = gn_slabs(N, HW, C, G);
    long stat_rows = (HW + S - 1) / S;
    const float* part = scratch;
    if (have_pre) {
        // one (mean, M2) summary per tile of pre->rows pixels: no statistics pass over x.  Many tiles per image are
        // merged once, by a small kernel, instead of by every apply block.
        part = pre->part; S = pre->S; stat_rows = pre->rows;
        if (S > 64) {
            hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(256), 0, s, part, scratch, S, G, stat_rows, HW, C / G);
            SD_HIP_CHECK(hipGetLastError());
            part = scratch; S = 1; stat_rows = HW;
        }
    } else {
        const int rc = launch_stats_pass(x, ldx, N, HW, C, G, S, scratch, s);
        if (rc) return rc;
    }
    const int CB = gn_block_channels(C, G);
    if (!old && CB / (C / G) <= 128) {
        // channel-blocked apply: rows per block = (256 / chunks per row) x NV, NV the largest of 8 / 4 / 2 / 1 that still
        // leaves about 512 blocks (every block pays the summary prologue: fewer, longer blocks won the sweep) (or one pixel row group per block on the small maps)
        const int cblocks = cdiv(C, CB);
        const int rows_par = 256 / (CB / 8) > 0 ? 256 / (CB / 8) : 1;
        int nv = 8;
        while (nv > 1 && cdiv(HW, (long)rows_par * nv) * cblocks * N < 512) nv >>= 1;
        const dim3 grid((unsigned)cdiv(HW, (long)rows_par * nv), cblocks, N);
#define SD_GN_APPLY2(NVV) hipLaunchKernelGGL((gn_apply2_kernel<NVV>), grid, dim3(256), 0, s, x, ldx, part, gamma, beta, y, ldy, HW, C, G, S, stat_rows, CB, eps, silu)
        if (nv == 8) SD_GN_APPLY2(8); else if (nv == 4) SD_GN_APPLY2(4); else if (nv == 2) SD_GN_APPLY2(2); else SD_GN_APPLY2(1);
#undef SD_GN_APPLY2
        SD_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // apply: rows per block so that a thread holds <= GN_APPLY_NV chunks, and >= ~512 blocks overall
    const int CC = C / 8;
    long rows_per = (long)256 * GN_APPLY_NV / CC;
    if (rows_per < 1) { set_error("groupnorm: C too large for the apply kernel"); return 1; }
    const long want = cdiv(HW * N, 512);
    if (rows_per > want) rows_per = want > 0 ? want : 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)cdiv(HW, rows_per), N), dim3(256),
                       ((size_t)C * 2 + (size_t)G * 2 + 768) * sizeof(float), s,
                       x, ldx, part, gamma, beta, y, ldy, HW, C, G, S, stat_rows, (int)rows_per, eps, silu);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

// GroupNorm over a channel concatenation [A | B] (the up blocks' torch.cat([hidden, skip])) from the summaries the two
// producers left: A's over Ga sub-groups of its Ca channels, B's over Gb of its Cb, every group of the concatenation a
// union of whole sub-groups (gn_cat_unit).  One block per image, 32 lanes per group; the merge is the pivoted
// weighted form of gn_apply2_kernel's: with p = the first summary's mean and n_k its element count,
//   a = sum n_k (m_k - p),  b = sum n_k (m_k - p)^2,  q = sum M2_k,  n = sum n_k:  mean = p + a / n,  M2 = q + b - a^2 / n.
// Output: one (mean, M2) per (image, group) = GnStats{out, 1, HW}; the apply pass then has one summary per group to read.
struct GnCatSrc { const float* part; int S; long rows; int G; int C; };
__global__ __launch_bounds__(1024) void gn_cat_finalize_kernel(GnCatSrc A, GnCatSrc B, float* __restrict__ out, long HW, int G) {
    // 32 lanes per group, the summaries dealt round-robin; four loads per lane go out back to back (unconditional, on a
    // clamped item: a first version with eight lanes per group and a plain loop took one round trip per summary, 8.9 us)
    const int n = blockIdx.x, tid = threadIdx.x;
    const int g = tid >> 5, pi = tid & 31;
    const int cpg = (A.C + B.C) / G;
    const int ua = A.C / A.G, ub = B.C / B.G;
    const int gq = g < G ? g : G - 1;
    const int c0 = gq * cpg, c1 = c0 + cpg;
    // sub-group ranges of this group in either source
    const int ca1 = c1 < A.C ? c1 : A.C;
    const int a0 = c0 < A.C ? c0 / ua : 0, a1 = c0 < A.C ? (ca1 + ua - 1) / ua : 0;
    const int b0 = c1 > A.C ? ((c0 > A.C ? c0 : A.C) - A.C) / ub : 0, b1 = c1 > A.C ? (c1 - A.C + ub - 1) / ub : 0;
    const int wa = a1 - a0, wb = b1 - b0;
    const int na = wa * A.S, nt = na + wb * B.S;
    const float* pa = A.part + (long)n * A.S * A.G * 2;
    const float* pb = B.part + (long)n * B.S * B.G * 2;
    float pivot = 0.f;                 // item 0's mean: lane 0 of the group has it after the first batch of loads
    float sa = 0.f, sb = 0.f, sq = 0.f, sn = 0.f;
    for (int i0 = pi; i0 < nt + pi; i0 += 32 * 4) {      // (same trip count for all 32 lanes of a group: shuffles inside)
        float2 v[4];
        float cnt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ii = i0 + 32 * q;
            const int i = ii < nt ? ii : nt - 1;
            const bool isa = i < na;
            const int j = isa ? i : i - na;
            const int w = isa ? wa : wb;
            const int k = j / w, sg = (isa ? a0 : b0) + j - k * w;
            v[q] = *reinterpret_cast<const float2*>((isa ? pa : pb) + ((long)k * (isa ? A.G : B.G) + sg) * 2);
            const long rws = isa ? A.rows : B.rows;
            long rows = HW - (long)k * rws;
            if (rows > rws) rows = rws;
            cnt[q] = ii < nt ? (float)rows * (float)(isa ? ua : ub) : 0.f;
        }
        if (i0 == pi) pivot = __shfl(v[0].x, (tid & 32), 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = v[q].x - pivot;
            sa += cnt[q] * d; sb += cnt[q] * d * d; sq += cnt[q] > 0.f ? v[q].y : 0.f; sn += cnt[q];
        }
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); sq += __shfl_xor(sq, o, 64); sn += __shfl_xor(sn, o, 64);
    }
    if (pi == 0 && g < G) {
        const float dm = sa / sn;
        *reinterpret_cast<float2*>(out + ((long)n * G + g) * 2) = float2{pivot + dm, sq + sb - sa * dm};
    }
}

// Width of the sub-groups a producer of Ca of the concatenation's Ca + Cb channels must summarise so that every one of
// the G groups is a union of whole sub-groups (the seam included): gcd(channels per group, Ca).
int gn_cat_unit(int Ca, int Cb, int G) {
    int a = (Ca + Cb) / G, b = Ca;
    while (b) { const int t = a % b; a = b; b = t; }
    return a;
}

int launch_gn_cat_finalize(const GnStats& sa, int Ga, int Ca, const GnStats& sb, int Gb, int Cb, float* out, int N, long HW, int G,
                           hipStream_t s) {
    const int C = Ca + Cb;
    if (G < 1 || G > 32 || C % G != 0 || Ga < 1 || Gb < 1 || Ca % Ga != 0 || Cb % Gb != 0 || !sa.part || !sb.part) {
        set_error("gn_cat_finalize: bad arguments"); return 1;
    }
    const int cpg = C / G, ua = Ca / Ga, ub = Cb / Gb;
    if (gn_cat_unit(Ca, Cb, G) % ua != 0 || cpg % ub != 0 || Ca % ub != 0) { set_error("gn_cat_finalize: sub-groups straddle a group boundary"); return 1; }
    GnCatSrc A{sa.part, sa.S, sa.rows, Ga, Ca}, B{sb.part, sb.S, sb.rows, Gb, Cb};
    hipLaunchKernelGGL(gn_cat_finalize_kernel, dim3(N), dim3(1024), 0, s, A, B, out, HW, G);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gn_stats(const half_t* x, long ldx, int N, long HW, int C, int G, float* scratch, GnStats* st, hipStream_t s) {
    if (C % 8 != 0 || C % G != 0 || G > 256) { set_error("groupnorm: C must be a multiple of 8 and of groups"); return 1; }
    const int S = gn_slabs(N, HW, C, G);
    const int rc = launch_stats_pass(x, ldx, N, HW, C, G, S, scratch, s);
    if (rc) return rc;
    st->part = scratch; st->S = S; st->rows = (HW + S - 1) / S;
    return 0;
}

int launch_gn_finalize(GnStats* st, float* out, int N, long HW, int C, int G, hipStream_t s) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(256), 0, s, st->part, out, st->S, G, st->rows, HW, C / G);
    SD_HIP_CHECK(hipGetLastError());
    st->part = out; st->S = 1; st->rows = HW;
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

int launch_row_stats(const half_t* x, long ldx, float* stat, long rows, int C, hipStream_t s) {
    if (C % 8 != 0) { set_error("row_stats: C must be a multiple of 8"); return 1; }
    hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, stat, rows, C);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sd
