// The two ends of the UNet (gfx950 / CDNA4): conv_in (a handful of latent channels -> 320) and the tail
// norm_out -> SiLU -> conv_out (320 -> 4, NCHW).  diffusers' UNet2DConditionModel.forward, reached from the reference at
// /root/reference/pipelines/sd_unified_pipeline.py:475-482.
//
// Neither is a GEMM worth a GEMM kernel: conv_in has K = 36 and writes 21 MB (CFG batch 8 at 64 x 64), conv_out reads
// 21 MB and writes 0.26 MB -- both are bound by that one pass over the big tensor.  On the generic paths they cost
// 71 us (im2col kernel + 64-wide-K GEMM) and 62 us (GroupNorm apply + 64-column implicit GEMM that re-gathers x nine
// times through the DMA path + NHWC->NCHW) per forward; here each is one launch that touches the big tensor once:
//   conv_head_kernel   im2col tile built in LDS straight from the NCHW latents, [128 x 64] x [64 x 320] on the MFMAs,
//                      output tile through LDS (coalesced 16-byte stores), GroupNorm summaries of the tile from that
//                      LDS image for the first resnet's norm1 (IGemmParams::gnstat_out layout);
//   conv_tail_kernel   an 8 x 16 pixel tile with its halo (10 x 18 pixels x all channels, 115 KB) is normalised and
//                      SiLU'd on the way from global memory into LDS (summaries of the producer merged in the
//                      prologue), the 3 x 3 convolution runs on MFMAs with the 4 output channels padded to a 16-wide
//                      tile in registers (the other 12 lanes read a zero row), NCHW stores.
#include <cstdlib>

#include "kernels.h"

namespace sd {
namespace {

__device__ __forceinline__ void edge_chan_merge(float& nA, float& mA, float& qA, float nB, float mB, float qB) {
    if (nB <= 0.f) return;
    const float n = nA + nB;
    const float d = mB - mA;
    const float f = nB / n;
    mA += d * f;
    qA += qB + d * d * nA * f;
    nA = n;
}

// ------------------------------------------------------------------------------------------------------------------
// conv_in
// ------------------------------------------------------------------------------------------------------------------
constexpr int HD_BM = 128, HD_COUT = 320, HD_K = 64, HD_LDC = HD_COUT + 8;
struct HeadLds {
    static constexpr int A = HD_BM * HD_K * 2;            // 16384
    static constexpr int B = HD_COUT * HD_K * 2;          // 40960
    static constexpr int Cc = HD_BM * HD_LDC * 2;         // 83968
    static constexpr int TOTAL = A + B + Cc;              // 141312
};

__global__ __launch_bounds__(512) void conv_head_kernel(HeadParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sA = reinterpret_cast<half_t*>(smem);
    half_t* sB = reinterpret_cast<half_t*>(smem + HeadLds::A);
    half_t* sC = reinterpret_cast<half_t*>(smem + HeadLds::A + HeadLds::B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const long HW = (long)p.H * p.W;
    const long m0 = (long)blockIdx.x * HD_BM;
    const int img = (int)(m0 / HW);

    // ---- weights [320][64] -> LDS (chunk ^ (row & 7)) ----
#pragma unroll
    for (int it = 0; it < HD_COUT * 8 / 512; ++it) {
        const int idx = tid + it * 512, row = idx >> 3, cc = idx & 7;
        *reinterpret_cast<h8*>(sB + row * 64 + ((cc ^ (row & 7)) << 3)) = *reinterpret_cast<const h8*>(p.w + (long)row * HD_K + cc * 8);
    }
    // ---- the im2col tile: thread = (pixel, quarter of K); k = (kh * 3 + kw) * Cin + c, zero from 9 Cin up ----
    {
        const int px = tid & 127, q = tid >> 7;
        const long pm = m0 + px - (long)img * HW;
        const int y = (int)(pm / p.W), x = (int)(pm % p.W);
        const half_t* xb = p.x_nchw + (long)img * p.Cin * HW;
        const int kreal = 9 * p.Cin;
        h8 v[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = q * 16 + e;
            half_t val = (half_t)0.f;
            if (k < kreal) {
                const int tap = k / p.Cin, c = k - tap * p.Cin;
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) val = xb[(long)c * HW + (long)yy * p.W + xx];
            }
            v[e >> 3][e & 7] = val;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<h8*>(sA + px * 64 + (((q * 2 + h) ^ (px & 7)) << 3)) = v[h];
    }
    __syncthreads();

    // ---- [128 x 64] x [64 x 320]: 8 waves as 2 (rows) x 4 (columns), 64 x 80 each ----
    const int wm = wave >> 2, wn = wave & 3;
    f4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + fq) ^ (fr & 7)) << 3;
        h8 fa[4], fb[5];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const h8*>(sA + (wm * 64 + i * 16 + fr) * 64 + c);
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const h8*>(sB + (wn * 80 + j * 16 + fr) * 64 + c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    // ---- + bias, fp16, through LDS ----
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int col = wn * 80 + j * 16 + fq * 4;
        const f4 b = *reinterpret_cast<const f4*>(p.bias + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f4 v = acc[i][j] + b;
            h4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
            *reinterpret_cast<h4*>(sC + (wm * 64 + i * 16 + fr) * HD_LDC + col) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < HD_BM * (HD_COUT / 8) / 512; ++it) {
        const int idx = tid + it * 512, row = idx / (HD_COUT / 8), cc = idx - row * (HD_COUT / 8);
        *reinterpret_cast<h8*>(p.y + (m0 + row) * p.ldy + cc * 8) = *reinterpret_cast<const h8*>(sC + row * HD_LDC + cc * 8);
    }
    // ---- GroupNorm summaries of the tile as stored (fp16 values): (mean, M2) per group, two passes over the LDS image;
    //      16 lanes per group, 8 rows each ----
    if (p.gnstat_out) {
        const int cpg = HD_COUT / p.G;
        const int g = tid >> 4, sub = tid & 15;
        if (g < p.G) {
            float sm = 0.f;
            for (int r = 0; r < 8; ++r) {
                const half_t* row = sC + (sub * 8 + r) * HD_LDC + g * cpg;
                for (int c = 0; c < cpg; ++c) sm += (float)row[c];
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o, 64);
            const float mean = sm / (float)(HD_BM * cpg);
            float q = 0.f;
            for (int r = 0; r < 8; ++r) {
                const half_t* row = sC + (sub * 8 + r) * HD_LDC + g * cpg;
                for (int c = 0; c < cpg; ++c) { const float d = (float)row[c] - mean; q += d * d; }
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o, 64);
            if (sub == 0) {
                const long tile = (m0 - (long)img * HW) / HD_BM;
                *reinterpret_cast<float2*>(p.gnstat_out + (((long)img * (HW / HD_BM) + tile) * p.G + g) * 2) = float2{mean, q};
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------------------------
// norm_out -> SiLU -> conv_out
// ------------------------------------------------------------------------------------------------------------------
constexpr int TL_TR = 8, TL_TC = 16, TL_HR = TL_TR + 2, TL_HC = TL_TC + 2, TL_NPX = TL_HR * TL_HC;      // 180 halo pixels
constexpr int kTailMaxGroups = 32, kTailMaxParts = 16;

template <int C>
struct TailLds {
    static constexpr int X = (C / 64) * TL_NPX * 128;     // C = 320: 115200
    static constexpr int WLD = 9 * C + 8;                 // halves per weight row (+16 B: rows 0..4 on different banks)
    static constexpr int Wt = 5 * WLD * 2;                // rows 0-3 = output channels, row 4 = zeros: 28880
    static constexpr int COEF = C * 8;                    // (scale, shift) per channel
    static constexpr int RED = kTailMaxGroups * kTailMaxParts * 3 * 4 + kTailMaxGroups * 2 * 4;
    static constexpr int TOTAL = X + Wt + COEF + RED;
};

template <int C>
__global__ __launch_bounds__(512) void conv_tail_kernel(TailParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CH8 = C / 8, SLABS = C / 64, NPX = TL_NPX, WLD = TailLds<C>::WLD;
    constexpr int NIT = (NPX * CH8 + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sX = reinterpret_cast<half_t*>(smem);
    half_t* sW = reinterpret_cast<half_t*>(smem + TailLds<C>::X);
    float2* sCoef = reinterpret_cast<float2*>(smem + TailLds<C>::X + TailLds<C>::Wt);
    float* sRed = reinterpret_cast<float*>(smem + TailLds<C>::X + TailLds<C>::Wt + TailLds<C>::COEF);
    float* sGS = sRed + kTailMaxGroups * kTailMaxParts * 3;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tilesX = p.W / TL_TC, tilesY = p.H / TL_TR;
    const int n = blockIdx.x / (tilesX * tilesY);
    const int t = blockIdx.x - n * tilesX * tilesY;
    const int ty = t / tilesX, tx = t - ty * tilesX;
    const long HW = (long)p.H * p.W;

    // ---- the halo tile, raw, into registers (its round trip overlaps the prologue below) ----
    h8 xv[NIT];
    bool inimg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 512;
        const int px = idx / CH8, ch = idx - px * CH8;
        const int hy = px / TL_HC, hx = px - hy * TL_HC;
        const int gy = ty * TL_TR + hy - 1, gx = tx * TL_TC + hx - 1;
        inimg[it] = idx < NPX * CH8 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        xv[it] = inimg[it] ? *reinterpret_cast<const h8*>(p.x + ((long)n * HW + (long)gy * p.W + gx) * p.ldx + ch * 8)
                           : h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    // ---- weights: rows < Cout from the packed matrix, the rest zero ----
    for (int idx = tid; idx < 5 * (9 * C / 8); idx += 512) {
        const int r = idx / (9 * C / 8), cc = idx - r * (9 * C / 8);
        *reinterpret_cast<h8*>(sW + r * WLD + cc * 8) =
            r < p.Cout ? *reinterpret_cast<const h8*>(p.w + (long)r * p.K + cc * 8) : h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    // ---- GroupNorm: the producer's summaries of image n merged (fixed order), then (scale, shift) per channel ----
    const int cpg = C / p.G;
    {
        const int gi = tid % p.G, pi = tid / p.G;
        if (pi < kTailMaxParts) {
            float nA = 0.f, mA = 0.f, qA = 0.f;
            const float* src = p.gn_part + ((long)n * p.gn_S * p.G + gi) * 2;
            for (int k = pi; k < p.gn_S; k += kTailMaxParts) {
                long rows = HW - (long)k * p.gn_rows;
                if (rows > p.gn_rows) rows = p.gn_rows;
                edge_chan_merge(nA, mA, qA, (float)rows * (float)cpg, src[(long)k * p.G * 2], src[(long)k * p.G * 2 + 1]);
            }
            sRed[(pi * p.G + gi) * 3] = nA; sRed[(pi * p.G + gi) * 3 + 1] = mA; sRed[(pi * p.G + gi) * 3 + 2] = qA;
        }
    }
    __syncthreads();
    if (tid < p.G) {
        float nA = sRed[tid * 3], mA = sRed[tid * 3 + 1], qA = sRed[tid * 3 + 2];
        for (int k = 1; k < kTailMaxParts; ++k) edge_chan_merge(nA, mA, qA, sRed[(k * p.G + tid) * 3], sRed[(k * p.G + tid) * 3 + 1], sRed[(k * p.G + tid) * 3 + 2]);
        const float var = qA / ((float)HW * (float)cpg);
        sGS[tid * 2] = mA;
        sGS[tid * 2 + 1] = rsqrtf((var < 0.f ? 0.f : var) + p.eps);
    }
    __syncthreads();
    if (tid < C) {
        const int g = tid / cpg;
        const float w = sGS[g * 2 + 1] * p.gamma[tid];
        sCoef[tid] = float2{w, p.beta[tid] - sGS[g * 2] * w};
    }
    __syncthreads();
    // ---- normalise + SiLU on the way into LDS: [64-channel slab][halo pixel][128 B], chunk ^ (pixel & 7); the zero
    //      padding of the convolution is zero AFTER the norm, so out-of-image pixels are written as zeros ----
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 512;
        if (idx < NPX * CH8) {
            const int px = idx / CH8, ch = idx - px * CH8;
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float2 cf = sCoef[ch * 8 + e];
                float v = __builtin_fmaf((float)xv[it][e], cf.x, cf.y);
                if (p.silu) v = silu_f(v);
                o[e] = inimg[it] ? (half_t)v : (half_t)0.f;
            }
            *reinterpret_cast<h8*>(sX + ((ch >> 3) * NPX + px) * 64 + (((ch & 7) ^ (px & 7)) << 3)) = o;
        }
    }
    __syncthreads();

    // ---- 3 x 3 convolution: wave = one row of 16 pixels; 9 taps x C / 32 MFMAs, one accumulator per kernel row ----
    f4 acc[3];
    const int wr = fr < 4 ? fr : 4;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        acc[kh] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int pa = (wave + kh) * TL_HC + fr + kw;
            const half_t* xa = sX + pa * 64;
            const half_t* wa = sW + wr * WLD + (kh * 3 + kw) * 64 + fq * 8;       // packed K order [C / 64][kh][kw][64]
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const h8 fa = *reinterpret_cast<const h8*>(xa + sl * NPX * 64 + (((ks * 4 + fq) ^ (pa & 7)) << 3));
                    const h8 fb = *reinterpret_cast<const h8*>(wa + sl * 9 * 64 + ks * 32);
                    acc[kh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, fa, acc[kh], 0, 0, 0);
                }
        }
    }
    if (fq == 0) {
        const f4 v = acc[0] + acc[1] + acc[2];
        const int gy = ty * TL_TR + wave, gx = tx * TL_TC + fr;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < p.Cout) p.y[(((long)n * p.Cout + e) * p.H + gy) * p.W + gx] = (half_t)(v[e] + p.bias[e]);
    }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

bool conv_head_supported(const HeadParams& p) {
    static const bool off = getenv("SD_NO_EDGE_KERNELS") != nullptr;
    const long HW = (long)p.H * p.W;
    return !off && p.Cout == HD_COUT && p.K == HD_K && 9 * p.Cin <= HD_K && HW % HD_BM == 0 &&
           (!p.gnstat_out || (p.G >= 1 && p.G <= 32 && HD_COUT % p.G == 0)) && p.ldy % 8 == 0;
}

int launch_conv_head(const HeadParams& p, hipStream_t s) {
    if (!conv_head_supported(p)) { set_error("conv_head: unsupported problem"); return 1; }
    static_assert(HeadLds::TOTAL <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    if (attr_once.first())
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HeadLds::TOTAL));
    const long M = (long)p.N * p.H * p.W;
    hipLaunchKernelGGL(conv_head_kernel, dim3((unsigned)(M / HD_BM)), dim3(512), HeadLds::TOTAL, s, p);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

bool conv_tail_supported(const TailParams& p) {
    static const bool off = getenv("SD_NO_EDGE_KERNELS") != nullptr;
    return !off && p.C == 320 && p.Cout >= 1 && p.Cout <= 4 && p.K == 9L * p.C && p.H % TL_TR == 0 && p.W % TL_TC == 0 &&
           p.G >= 1 && p.G <= kTailMaxGroups && p.C % p.G == 0 && p.gn_part && p.gn_S >= 1 && p.ldx % 8 == 0;
}

int launch_conv_tail(const TailParams& p, hipStream_t s) {
    if (!conv_tail_supported(p)) { set_error("conv_tail: unsupported problem"); return 1; }
    static_assert(TailLds<320>::TOTAL <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    if (attr_once.first())
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tail_kernel<320>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         TailLds<320>::TOTAL));
    const unsigned blocks = (unsigned)((long)p.N * (p.H / TL_TR) * (p.W / TL_TC));
    hipLaunchKernelGGL(conv_tail_kernel<320>, dim3(blocks), dim3(512), TailLds<320>::TOTAL, s, p);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sd
