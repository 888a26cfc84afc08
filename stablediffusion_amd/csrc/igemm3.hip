// Pointwise GEMM with the ACTIVATION operand on the ordinary vector-memory path (gfx950 / CDNA4):
//     y[m, n] = bias[n] + res[m, n] + sum_k x[m, k] W[n, k]       (+ LayerNorm consumer form, + row statistics)
// the C x C / C x 4C linears of the transformer blocks at the 32 x 32, 16 x 16 and 8 x 8 levels (to_q, to_out, proj_in,
// proj_out, ff2; diffusers BasicTransformerBlock / Transformer2DModel under
// /root/reference/pipelines/sd_unified_pipeline.py:475-482).
//
// Why: tools/probe_dma.py (sd_probe_lds_dma).  With every CU streaming, the LDS-DMA path (buffer_load ... lds) delivers
// 8-10 TB/s chip-wide from L2 however deep the ring -- ordinary 16-byte loads into registers deliver 30 TB/s from the same
// L2.  igemm2_kernel brings BOTH operands through the DMA path and its small-tile launches sit at 6.3-7.7 TB/s of it: the
// path, not the matrix pipe (3 us of MFMA in a 20 us launch), not the ring depth, is their bound.  In the 128 x 80 tile with
// four waves stacked along M no two waves share an activation row, so the activation fragments need no LDS at all: each
// lane fetches its MFMA operand (row fr of the wave's 32, 16 bytes at the step's k) straight from global memory into
// registers, a ring of STAGES slabs deep, D of them in flight.  Only the weights (10 KB per slab instead of 26 KB) use the
// DMA path and the LDS; a block needs 42 KB of it instead of 78-104 KB.
//
// Same tile order (XCD remap, mfast), weight layout, epilogue arithmetic and statistics layout as igemm2_kernel<128, 80,
// 4, 1, .>.  MEASURED: not faster -- 15.8 vs 15.3 us on 2048 x 1280 x 1280, 16.5 vs 12.8 on 8192 x 640 x 640, 48 vs 33 on
// 8192 x 640 x 2560 (profiles/r03_lds_dma_probe.txt): moving the activation operand off the DMA path did not lift these
// launches, so the path's instruction rate is not what holds them; they sit at ~8.6 TB/s of combined operand traffic
// either way.  Kept as variant 18 (sd_igemm_force(18, 1), SD_IGEMM3=1 routes the 128 x 80 launches to it) with its parity
// test, as the base for the next attempt (a larger tile with only the weights in LDS).
#include <cstdlib>

#include "kernels.h"

namespace sd {
namespace {

constexpr int G3_BM = 128, G3_BN = 80, G3_STAGES = 4;
constexpr int kG3MaxLnParts = 20;

template <int N>
__device__ __forceinline__ void g3_wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int PER, int YMAX>
__device__ __forceinline__ void g3_wait_slabs(int y) {
    if constexpr (YMAX == 0) {
        g3_wait_vmcnt<0>();
    } else {
        if (y >= YMAX) g3_wait_vmcnt<PER * YMAX>();
        else g3_wait_slabs<PER, YMAX - 1>(y);
    }
}

struct G3Lds {
    static constexpr int RING = G3_STAGES * G3_BN * 64 * 2;          // weights only: 4 x 10 KB
    static constexpr int EPI = G3_BM * (G3_BN + 8) * 2;              // output tile staging
    static constexpr int RED = G3_BM * (G3_BN / 8) * 8;              // row-statistics chunk partials
    static constexpr int MAIN = RING > EPI + RED ? RING : EPI + RED;
    static constexpr int TOTAL = MAIN + G3_BM * 8;                   // + (mean, rstd) of the block's rows
};

__global__ __launch_bounds__(256) void igemm3_kernel(IGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = G3_BM, BN = G3_BN, STAGES = G3_STAGES, D = STAGES - 1;
    constexpr int NW = 4, NT = 256, TM = 2, TN = 5, LDC = BN + 8;
    constexpr int B_INSTR = BN / 8, B_REM = B_INSTR % NW, B_PW = (B_INSTR + NW - 1) / NW;      // 10 pieces: 3, 3, 2, 2
    constexpr int A_LD = TM * 2;                 // activation fragment loads per lane per slab
    constexpr int PER_HI = B_PW + A_LD, PER_LO = B_PW - 1 + A_LD;
    constexpr int STAGE_HALVES = BN * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* ring = reinterpret_cast<half_t*>(smem);
    half_t* sC = reinterpret_cast<half_t*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + G3Lds::EPI);
    float* sStat = reinterpret_cast<float*>(smem + G3Lds::MAIN);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (p.mfast) {
        const int tiles_m = gridDim.x / tiles_n;
        tn = bid / tiles_m; tm = bid - tn * tiles_m;
    } else {
        tm = bid / tiles_n; tn = bid - tm * tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K / 64;

    // ---- descriptors: x range-checked at M rows (rows past M read as zeros), W at its padded rows ----
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, (int)((long)p.M * p.ldx * 2), 0x00020000);
    const long wrows = ((long)p.Cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, (int)(wrows * p.K * 2), 0x00020000);
    constexpr unsigned kOOB = 0x80000000u;
    // activation fragments: lane (fr, fq) of wave w holds row m0 + 32 w + 16 i + fr, halves [64 kt + 32 ks + 8 fq, + 8)
    unsigned a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long m = m0 + wave * 32 + i * 16 + fr;
        a_off[i] = m < p.M ? (unsigned)((m * p.ldx + fq * 8) * 2) : kOOB;
    }
    // weight slab [80 rows][64 k] by DMA, chunk ^ (row & 7) on the source side
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (lrow & 7);
    const bool b_hi = wave < B_REM;                                   // wave-uniform: three pieces, else two
    const int b_cnt = b_hi ? B_PW : B_PW - 1;
    const int b_first = b_hi ? wave * B_PW : B_REM * B_PW + (wave - B_REM) * (B_PW - 1);
    unsigned b_off[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j) b_off[j] = (unsigned)((((long)(n0 + (b_first + j) * 8 + lrow)) * p.K + chunk * 8) * 2);

    auto issue_b = [&](int slot, int kt) {
        half_t* sb = ring + slot * STAGE_HALVES;
#pragma unroll
        for (int j = 0; j < B_PW; ++j)
            if (j < b_cnt)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sb + (b_first + j) * 512), 16,
                                                         b_off[j] + (unsigned)(kt * 128), 0, 0, 0);
    };
    // The activation loads are inline asm on purpose.  Through the builtin the compiler tracks them itself and, at the
    // loop's back edge, no longer knows how many younger operations follow a fragment's load: it put s_waitcnt vmcnt(3..0)
    // in front of the step's MFMAs, i.e. drained every younger slab each step.  As asm the loads are invisible to its
    // counter; the counted waits below are tied to the fragment registers ("+v") so that no use can move above them.
    typedef int i4v __attribute__((ext_vector_type(4)));
    const unsigned long long xa = (unsigned long long)p.x;
    i4v dx;
    dx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xa);
    dx[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xa >> 32) & 0xffffu));
    dx[2] = __builtin_amdgcn_readfirstlane((int)((long)p.M * p.ldx * 2));
    dx[3] = 0x00020000;
    (void)rx;
    h8 ar[STAGES][TM][2];
    auto load_a = [&](int slot, int kt) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned voff = a_off[i] + (unsigned)(kt * 128 + ks * 64);      // (kOOB + a few KB stays out of range)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ar[slot][i][ks]) : "v"(voff), "s"(dx) : "memory");
            }
    };
    auto tie_a = [&](int slot) {
        asm volatile("" : "+v"(ar[slot][0][0]), "+v"(ar[slot][0][1]), "+v"(ar[slot][1][0]), "+v"(ar[slot][1][1]));
    };

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: slabs 0 .. D-1, weights then activations of each (the order the waits below count on) ----
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nk) { issue_b(s, s); load_a(s, s); }

    // LayerNorm consumer: mean / rstd of the block's rows from the producer's partial sums (igemm2_kernel's form)
    if (p.ln_stat) {
        if (tid < BM) {
            const int m = m0 + tid;
            float2 pv[kG3MaxLnParts];
            const float2* src = reinterpret_cast<const float2*>(p.ln_stat) + (long)(m < p.M ? m : 0) * p.ln_parts;
#pragma unroll
            for (int k = 0; k < kG3MaxLnParts; ++k) pv[k] = src[k < p.ln_parts ? k : p.ln_parts - 1];
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int k = 0; k < kG3MaxLnParts; ++k) {
                sm += k < p.ln_parts ? pv[k].x : 0.f;
                sq += k < p.ln_parts ? pv[k].y : 0.f;
            }
            const float inv = 1.0f / (float)p.ln_C;
            const float mean = sm * inv;
            float var = sq * inv - mean * mean;
            var = var < 0.f ? 0.f : var;
            sStat[tid * 2] = mean;
            sStat[tid * 2 + 1] = rsqrtf(var + p.ln_eps);
        }
    }

    for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
        for (int slot = 0; slot < STAGES; ++slot) {
            const int kt = kt0 + slot;
            if (kt >= nk) break;
            // slab kt (weights AND this wave's activation fragments) has landed; the younger slabs -- (pieces + 4 loads)
            // each, issued in that order -- stay in flight
            {
                const int rem = nk - 1 - kt;
                const int y = rem < D - 1 ? rem : D - 1;
                if (b_hi) g3_wait_slabs<PER_HI, D - 1>(y); else g3_wait_slabs<PER_LO, D - 1>(y);
            }
            tie_a(slot);
            __builtin_amdgcn_s_barrier();
            const int pslot = slot == 0 ? STAGES - 1 : slot - 1;          // (kt + D) % STAGES
            if (kt + D < nk) issue_b(pslot, kt + D);
            const half_t* cB = ring + slot * STAGE_HALVES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 fb[TN];
                const int ch = ks * 4 + fq;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = j * 16 + fr;
                    fb[j] = *reinterpret_cast<const h8*>(cB + r * 64 + ((ch ^ (r & 7)) << 3));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], ar[slot][i][ks], acc[i][j], 0, 0, 0);
            }
            // the register set of slab kt - 1 is free again: slab kt + D goes there
            if (kt + D < nk) load_a(pslot, kt + D);
        }
    }

    // ---- epilogue (as igemm2_kernel): accumulators -> LDS (LayerNorm correction, bias) -> output pass (residual, stores,
    //      row statistics) ----
    __syncthreads();
    f4 bias4[TN], wsum4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        bias4[j] = p.bias ? *reinterpret_cast<const f4*>(p.bias + n0 + j * 16 + fq * 4) : f4{0.f, 0.f, 0.f, 0.f};
    if (p.ln_stat) {
#pragma unroll
        for (int j = 0; j < TN; ++j) wsum4[j] = *reinterpret_cast<const f4*>(p.ln_wsum + n0 + j * 16 + fq * 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pr = wave * 32 + i * 16 + fr;
        if (p.ln_stat) {
            const float mean = sStat[pr * 2], rstd = sStat[pr * 2 + 1];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (acc[i][j] - mean * wsum4[j]) * rstd;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f4 v = acc[i][j] + bias4[j];
            h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<h4*>(sC + pr * LDC + j * 16 + fq * 4) = o;
        }
    }
    __syncthreads();
    constexpr int CH = BN / 8, RPP = NT / CH, ITER = (BM + RPP - 1) / RPP;        // 10 chunks, 25 rows per pass, 6 passes
    const int c8 = tid % CH, rr = tid / CH;
    const int c = c8 * 8, n = n0 + c;
    const bool okc = rr < RPP && n < p.Cout;
    h8 rv[ITER];
    if (p.res) {
        const int nc = n < p.Cout ? n : 0;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int r = rr + it * RPP;
            int m = m0 + (r < BM ? r : BM - 1);
            m = m < p.M ? m : p.M - 1;
            rv[it] = *reinterpret_cast<const h8*>(p.res + (long)m * p.ldres + nc);
        }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int r = rr + it * RPP;
        const int m = m0 + r;
        if (okc && r < BM && m < p.M) {
            h8 v = *reinterpret_cast<const h8*>(sC + r * LDC + c);
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[it][e]);
            }
            *reinterpret_cast<h8*>(p.y + (long)m * p.ldy + n) = v;
            if (p.rowstat_out) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sm += f; sq += f * f; }
                *reinterpret_cast<float2*>(sRed + (r * CH + c8) * 2) = float2{sm, sq};
            }
        }
    }
    if (p.rowstat_out) {
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            const int nch = (p.Cout - n0 < BN ? p.Cout - n0 : BN) / 8;
            float sm = 0.f, sq = 0.f;
            for (int k = 0; k < nch; ++k) { const float2 v = *reinterpret_cast<const float2*>(sRed + (tid * CH + k) * 2); sm += v.x; sq += v.y; }
            *reinterpret_cast<float2*>(p.rowstat_out + ((long)(m0 + tid) * p.rowstat_parts + tn) * 2) = float2{sm, sq};
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

bool igemm3_supported(const IGemmParams& p) {
    if (!(p.KS == 1 && p.stride == 1 && p.up == 0) || p.geglu || p.act || p.rowadd || p.gnstat_out || p.gni_part) return false;
    if (p.acc_scale != 1.f || p.bias_scale != 1.f) return false;
    if (p.K % 64 != 0 || p.K < 64 * G3_STAGES || p.Cout % 8 != 0) return false;
    if (p.ln_stat && p.ln_parts > kG3MaxLnParts) return false;
    if ((long)p.M * p.ldx * 2 >= (1L << 31)) return false;
    const long wrows = ((long)p.Cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad;
    return wrows * p.K * 2 < (1L << 31);
}

int launch_igemm3(const IGemmParams& p, int mfast, hipStream_t s) {
    if (!igemm3_supported(p)) { set_error("igemm3: unsupported problem"); return 1; }
    static_assert(G3Lds::TOTAL <= 64 * 1024, "three blocks per CU");
    static PerDeviceOnce attr_once;
    if (attr_once.first())
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G3Lds::TOTAL));
    IGemmParams q = p;
    q.mfast = mfast;
    q.rowstat_parts = cdiv(p.Cout, G3_BN);
    const int tiles = cdiv(p.M, G3_BM) * cdiv(p.Cout, G3_BN);
    hipLaunchKernelGGL(igemm3_kernel, dim3(tiles), dim3(256), G3Lds::TOTAL, s, q);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sd
