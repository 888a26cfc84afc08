// Fused GEGLU feed-forward of the 64 x 64 level (gfx950 / CDNA4):
//     out = x + GEGLU(LN(x) W1^T + b1) W2^T + b2,     GEGLU(p) = p[:, :4C] * gelu(p[:, 4C:]),   C = 320
// diffusers' FeedForward under BasicTransformerBlock (norm3 -> ff -> + residual), reached from the reference at
// /root/reference/pipelines/sd_unified_pipeline.py:475-482.
//
// Unfused (igemm2 / wsgemm: ff1 + GEGLU, then ff2 + residual) the 4C-wide hidden tensor makes a round trip through
// memory: 84 MB written and 84-168 MB read back per block at CFG batch 8 (32768 x 1280 fp16; rocprofv3 FETCH_SIZE: 116 MB
// per ff1 launch alone), five times per forward, and ff2 (K = 1280, 256 tiles, one per CU) runs at the rate that read
// arrives.  Here the hidden tensor never leaves the CU:
//   * one block = 128 rows of x; eight waves as 4 (rows) x 2 (columns), and every wave keeps ITS 32 rows of x as MFMA
//     operand fragments in registers for the whole launch (80 VGPRs): the first GEMM's activation operand never touches
//     LDS, which leaves the LDS to a SEVEN-stage weight ring (6 slabs = 96-120 KB in flight per CU);
//   * the hidden dimension is walked in 20 steps of 64 units: S = x W1_step^T (128 x 128: 64 hidden + their 64 gates),
//     h = hidden * gelu(gate) -> LDS (16 KB), out += h W2_step^T (128 x 320, accumulated in registers over all steps);
//   * ONE stream of weight slabs (XOR-swizzled like every LDS-DMA tile here) feeds both GEMMs: per step five W1 slabs
//     [128 rows][64 k] and two W2 slabs [2 x 80 output columns][64 k] (a wave column owns 160 output columns, every slab
//     brings 80 of each so all eight waves work on every slab);
//   * LayerNorm folded as everywhere (statistics of the block's rows once, in the prologue), bias | wsum of a step by two
//     half pieces (waves 0 and 1), waits counted: every wave issues >= 2 pieces per slab, "at most 2 x 5 outstanding"
//     therefore means the slab about to be read has landed.
// All 256 blocks stream the same 2.46 MB of weights: they sit in every XCD's L2.
// Measured (profiles/r03_ffn_fused.txt): 114-116 us at M = 32768 against 127 us for the two GEMMs; with parts removed
// (results wrong, timing only): no GELU 98, no weight traffic 95, neither 77, no MFMA 70 -- the floor is the per-slab
// wait + barrier + DMA issue, not the LDS or the matrix pipe (32 us of MFMA).  The first version kept x in LDS (80 KB)
// and had room for a 3-stage ring only: 146 us.
#include <cstdlib>

#include "kernels.h"

namespace sd {
namespace {

constexpr int FF_C = 320, FF_BM = 128, FF_NKS = FF_C / 64, FF_NOS = 2, FF_STAGES = 7;
constexpr int kFfMaxLnParts = 20;

template <int N>
__device__ __forceinline__ void ff_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct FfLds {
    static constexpr int STAGE = 160 * 64 * 2;                   // 20480: a W2 slab (160 rows); a W1 slab uses 128 of them
    static constexpr int RING = FF_STAGES * STAGE;               // 143360
    static constexpr int H = FF_BM * 64 * 2;                     // 16384: GEGLU output of one step
    static constexpr int AUX = 2 * 256 * 4;                      // [2][bias 128 | wsum 128]
    static constexpr int STAT = FF_BM * 2 * 4;                   // (mean, rstd) per row
    static constexpr int TOTAL = RING + H + AUX + STAT;          // 162816
};

__global__ __launch_bounds__(512) void ffn_fused_kernel(FfnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int BM = FF_BM, C = FF_C, NKS = FF_NKS, NOS = FF_NOS, STAGES = FF_STAGES, LA = STAGES - 1;
    constexpr int NW = 8, TM = 2;
    constexpr int STAGE_HALVES = FfLds::STAGE / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* ring = reinterpret_cast<half_t*>(smem);
    half_t* sH = reinterpret_cast<half_t*>(smem + FfLds::RING);
    float* sAux = reinterpret_cast<float*>(smem + FfLds::RING + FfLds::H);
    float* sStat = reinterpret_cast<float*>(smem + FfLds::RING + FfLds::H + FfLds::AUX);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // 4 x 2 waves: 32 rows x 64 columns of a 128 x 128 product
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int NSTEP = p.hidden / 64;                          // 20

    // ---- x: this wave's 32 rows as MFMA operand fragments, in registers for the whole launch (the first GEMM's
    //      activation operand never touches LDS: the LDS is the weight ring) ----
    h8 xf[TM][2 * NKS];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const half_t* xr = p.x + (long)(m0 + wm * 32 + i * 16 + fr) * p.ldx + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 2 * NKS; ++ks) xf[i][ks] = *reinterpret_cast<const h8*>(xr + ks * 32);
    }
    // ---- LayerNorm statistics of the block's rows (plain loads: nothing else is in flight yet) ----
    if (tid < BM) {
        float2 pv[kFfMaxLnParts];
        const float2* src = reinterpret_cast<const float2*>(p.ln_stat) + (long)(m0 + tid) * p.ln_parts;
#pragma unroll
        for (int k = 0; k < kFfMaxLnParts; ++k) pv[k] = src[k < p.ln_parts ? k : p.ln_parts - 1];
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int k = 0; k < kFfMaxLnParts; ++k) {
            sm += k < p.ln_parts ? pv[k].x : 0.f;
            sq += k < p.ln_parts ? pv[k].y : 0.f;
        }
        const float inv = 1.0f / (float)C;
        const float mean = sm * inv;
        float var = sq * inv - mean * mean;
        var = var < 0.f ? 0.f : var;
        sStat[tid * 2] = mean;
        sStat[tid * 2 + 1] = rsqrtf(var + p.ln_eps);
    }
    __syncthreads();

    // ---- descriptors; a DMA piece = 8 tile rows x 128 B, chunk ^ (row & 7) on the SOURCE side ----
    __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w1), 0, (int)((long)p.w1_rows * C * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w2), 0, (int)((long)p.w2_rows * p.hidden * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wave == 0 ? p.b1 : p.wsum1), 0, p.w1_rows * 4, 0x00020000);
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (lrow & 7);
    // W1 slab, LDS rows [hidden 0-31 | gate 0-31 | hidden 32-63 | gate 32-63] of the packed [64 hidden | 64 gate] group:
    // a wave column's 64 columns are 32 hidden units next to their own gates (as igemm2_kernel's GEGLU form)
    auto gperm = [](int r) { return r < 32 || r >= 96 ? r : (r < 64 ? r + 32 : r - 32); };
    // W2 slab s, LDS row r: wave column r / 80 owns output columns [160 wn, 160 wn + 160); the slab holds 80 of them
    unsigned w1_off[2], w2_off[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 8 + lrow;
        w1_off[j] = (unsigned)((((long)gperm(r)) * C + chunk * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = (wave + NW * j) * 8 + lrow;                    // pieces wave, wave + 8, wave + 16 (< 20)
        const int oc = 160 * (r / 80) + (r % 80);
        w2_off[j] = (unsigned)((((long)oc) * p.hidden + chunk * 8) * 2);
    }

    // ---- the slab stream: step `it` = [W1(it) x NKS] then [W2(it) x NOS]; state of the next slab to issue ----
    int s_it = 0, s_q = 0, s_slot = 0;
    auto issue = [&]() {
        half_t* dst = ring + s_slot * STAGE_HALVES;
        const bool live = s_it < NSTEP;              // the tail of the stream is out-of-range pieces (they only count)
        if (s_q < NKS) {
            if (s_q == 0 && wave < 2 && live && lane < 32)
                // bias (wave 0) | wsum (wave 1) of step s_it, read by its GEGLU after its first GEMM: half a piece each
                __builtin_amdgcn_raw_ptr_buffer_load_lds(raux, (__attribute__((address_space(3))) void*)(sAux + (s_it & 1) * 256 + wave * 128),
                                                         16, (unsigned)((s_it * 128 + lane * 4) * 4), 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw1, (__attribute__((address_space(3))) void*)(dst + (wave * 2 + j) * 512), 16,
                                                         live ? w1_off[j] + (unsigned)((s_it * 128 * C + s_q * 64) * 2) : kOOB, 0, 0, 0);
        } else {
            const unsigned sb = (unsigned)((((long)(s_q - NKS) * 80) * p.hidden + s_it * 64) * 2);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (wave + NW * j < 20)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void*)(dst + (wave + NW * j) * 512), 16,
                                                             live ? w2_off[j] + sb : kOOB, 0, 0, 0);
        }
        if (++s_q == NKS + NOS) { s_q = 0; ++s_it; }
        if (++s_slot == STAGES) s_slot = 0;
    };
#pragma unroll
    for (int s = 0; s < LA; ++s) issue();

    f4 acc1[TM][4], acc2[NOS][TM][5];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NOS; ++s)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc2[s][i][j] = f4{0.f, 0.f, 0.f, 0.f};
    }

    // per-lane LDS read offset (halves) inside a 16-row fragment: row fr, k-chunk (ks * 4 + fq) ^ (fr & 7); the fragment's
    // first row (a multiple of 16, wave-uniform) is added as a scalar
    int lb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) lb[ks] = fr * 64 + (((ks * 4 + fq) ^ (fr & 7)) << 3);
    const int rowH = wm * 32 * 64, rowB1 = wn * 64 * 64, rowB2 = wn * 80 * 64;
    int c_slot = 0;
    // one slab: wait until it has landed (every wave issues at least two pieces per slab, so "at most 2 (LA - 1) pieces
    // outstanding" means slab g is in: the LA - 1 younger slabs stay in flight), barrier, issue the slab LA ahead
    auto slab_begin = [&]() -> const half_t* {
        ff_wait_vmcnt<2 * (LA - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes (h) are done before the barrier
        __builtin_amdgcn_s_barrier();
        issue();
        const half_t* cB = ring + c_slot * STAGE_HALVES;
        if (++c_slot == STAGES) c_slot = 0;
        return cB;
    };

    for (int it = 0; it < NSTEP; ++it) {
        {
            // ---- S(it) = x W1_it^T: five K slabs, the activation fragments from registers ----
#pragma unroll
            for (int q = 0; q < NKS; ++q) {
                const half_t* cB = slab_begin();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 fb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const h8*>(cB + rowB1 + j * 1024 + lb[ks]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], xf[i][q * 2 + ks], acc1[i][j], 0, 0, 0);
                }
            }
            // ---- h = hidden * gelu(gate), LayerNorm correction and bias first; written to sH as an operand slab ----
            const float* ax = sAux + (it & 1) * 256 + wn * 32 + fq * 4;
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const f4 bh = *reinterpret_cast<const f4*>(ax + jh * 16), bg = *reinterpret_cast<const f4*>(ax + 64 + jh * 16);
                const f4 wh = *reinterpret_cast<const f4*>(ax + 128 + jh * 16), wg = *reinterpret_cast<const f4*>(ax + 192 + jh * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float2 mr = *reinterpret_cast<const float2*>(sStat + (wm * 32 + i * 16 + fr) * 2);
                    const f4 hv = (acc1[i][jh] - mr.x * wh) * mr.y + bh;
                    const f4 gv = (acc1[i][jh + 2] - mr.x * wg) * mr.y + bg;
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)(hv[e] * gelu_erf_f(gv[e]));
                    const int pr = wm * 32 + i * 16 + fr;
                    const int k = wn * 32 + jh * 16 + fq * 4;           // hidden unit of the step
                    *reinterpret_cast<h4*>(sH + pr * 64 + (((k >> 3) ^ (pr & 7)) << 3) + (k & 7)) = o;
                    acc1[i][jh] = f4{0.f, 0.f, 0.f, 0.f};
                    acc1[i][jh + 2] = f4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        // ---- out += h W2_it^T: two slabs of 2 x 80 output columns (the barrier of the first orders the h writes) ----
        {
#pragma unroll
            for (int s = 0; s < NOS; ++s) {
                const half_t* cB = slab_begin();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 fa[TM], fb[5];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const h8*>(sH + rowH + i * 1024 + lb[ks]);
#pragma unroll
                    for (int j = 0; j < 5; ++j) fb[j] = *reinterpret_cast<const h8*>(cB + rowB2 + j * 1024 + lb[ks]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < 5; ++j)
                            acc2[s][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc2[s][i][j], 0, 0, 0);
                }
            }
        }
    }
    ff_wait_vmcnt<0>();          // the stream's tail (out-of-range pieces)

    // ---- epilogue: out = fp16(acc + b2) + x ----
#pragma unroll
    for (int s = 0; s < NOS; ++s) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int col = 160 * wn + 80 * s + j * 16 + fq * 4;
            const f4 b2 = *reinterpret_cast<const f4*>(p.b2 + col);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long m = m0 + wm * 32 + i * 16 + fr;
                const f4 v = acc2[s][i][j] + b2;
                const h4 xr = *reinterpret_cast<const h4*>(p.x + m * p.ldx + col);
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)(half_t)v[e] + (float)xr[e]);
                *reinterpret_cast<h4*>(p.y + m * p.ldy + col) = o;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

bool ffn_fused_supported(const FfnParams& p) {
    static const bool off = getenv("SD_NO_FFN_FUSE") != nullptr;
    return !off && p.C == FF_C && p.hidden == 4 * FF_C && p.M % FF_BM == 0 && p.M / FF_BM >= 64 && p.ln_stat && p.ln_parts >= 1 &&
           p.ln_parts <= kFfMaxLnParts && p.w1_rows >= 2 * p.hidden && p.w2_rows >= FF_C &&
           (long)p.M * p.ldx * 2 < (1L << 31) && p.b1 && p.b2 && p.wsum1;
}

int launch_ffn_fused(const FfnParams& p, hipStream_t s) {
    if (!ffn_fused_supported(p)) { set_error("ffn_fused: unsupported problem"); return 1; }
    static_assert(FfLds::TOTAL <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, FfLds::TOTAL));
    }
    hipLaunchKernelGGL(ffn_fused_kernel, dim3(p.M / FF_BM), dim3(512), FfLds::TOTAL, s, p);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sd
