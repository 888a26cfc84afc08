// Implicit-GEMM convolution / linear, LDS-DMA pipeline (gfx950 / CDNA4).
//
// Same contract as igemm.hip (IGemmParams), different data path -- the one the MI355X wants:
//   * both operand tiles go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging
//     VGPRs, no ds_write pass.  One wave-instruction fills 8 tile rows x 128 B; the XOR swizzle that
//     makes the ds_read_b128 fragment reads conflict-free is applied on the per-lane SOURCE address
//     (the LDS image of a DMA is lane-linear).
//   * the im2col gather, zero padding, stride 2, nearest-2x upsample and the ragged last M tile
//     are all address arithmetic on the DMA's per-lane byte offset: an out-of-image tap uses an
//     out-of-range buffer offset, for which the hardware writes zeros into LDS.
//   * ring of STAGES LDS buffers, prefetch distance STAGES-1, ONE raw s_barrier per 64-deep K slab
//     and a COUNTED s_waitcnt vmcnt(N) so the next slab's DMA stays in flight across the barrier.
//   * optional split-K (gridDim.y): every z-slice writes an fp32 partial slab; the epilogue kernel
//     splitk_epilogue_kernel reduces the slabs in a fixed order (deterministic) and applies
//     bias / time-embedding row add / residual / GEGLU.
// Tile variants (BM x BN, waves, stages) are chosen per shape by pick_variant().
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "kernels.h"

namespace sd {
namespace {

constexpr int BK = 64;
constexpr int kMaxLnParts = 20;     // row-statistics partials a LayerNorm consumer reads per row, at most
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Wait until at most `y` (wave-uniform, 0 <= y <= YMAX) ring stages of PER DMA instructions each are
// still in flight: the immediate has to be a literal, so the count is dispatched over a short chain.
template <int PER, int YMAX>
__device__ __forceinline__ void wait_stages(int y) {
    if constexpr (YMAX == 0) {
        wait_vmcnt<0>();
    } else {
        if (y >= YMAX) wait_vmcnt<PER * YMAX>();
        else wait_stages<PER, YMAX - 1>(y);
    }
}

// LDS layout shared by the kernel and its launcher: [ring | epilogue staging + row-statistics partials]
// overlaid, then the LayerNorm row statistics of a consumer launch (written in the prologue, read in
// the epilogue, so outside everything the main loop touches).
template <int BM, int BN, int STAGES, int BKT>
struct IGemm2Lds {
    static constexpr int RING = STAGES * (BM + BN) * BKT * 2;
    static constexpr int EPI = BM * (BN + 8) * 2;
    static constexpr int RED = BM * (BN / 8) * 8;        // row-statistics chunk partials
    static constexpr int GRED = 512 * 16;                // GroupNorm partials, 4 floats per thread
    static constexpr int MAIN = RING > EPI + RED + GRED ? RING : EPI + RED + GRED;
    static constexpr int TOTAL = MAIN + BM * 8;
};

// GroupNorm summaries of one output tile (all BM rows inside one image): every thread hands in the
// sums / sums of squares of its column chunk's channels, split over the (at most two) groups the chunk
// touches; one thread per group of the tile adds them in a fixed order and stores
// (mean, M2 = Q - S * mean) at dst[g * 2].  Tile-level plain sums (BM * cpg <= a few thousand values),
// merged across tiles with chan_merge by the consumer.
template <int BM, int BN, int NT>
__device__ __forceinline__ void gn_tile_stats(float* sG, int tid, float gs0, float gq0, float gs1, float gq1, int n0,
                                              int Cout, int cpg, int groups, float* dst) {
    constexpr int CH = BN / 8;
    constexpr int RPP = NT / CH;
    *reinterpret_cast<f4*>(sG + tid * 4) = f4{gs0, gq0, gs1, gq1};
    __syncthreads();
    const int cols = Cout - n0 < BN ? Cout - n0 : BN;
    const int ng = cols / cpg;                           // groups of this tile (tile columns start on a group)
    if (tid < ng) {
        const int g = n0 / cpg + tid;                    // global group
        const int cfirst = (tid * cpg) / 8, clast = ((tid + 1) * cpg - 1) / 8;
        float sm = 0.f, sq = 0.f;
        for (int c8 = cfirst; c8 <= clast; ++c8) {
            const int part = ((n0 + c8 * 8) / cpg == g) ? 0 : 2;     // the chunk's first group, or its second
            for (int rr = 0; rr < RPP; ++rr) {
                const float2 v = *reinterpret_cast<const float2*>(sG + (rr * CH + c8) * 4 + part);
                sm += v.x; sq += v.y;
            }
        }
        const float mean = sm / (float)(BM * cpg);
        *reinterpret_cast<float2*>(dst + (long)g * 2) = float2{mean, sq - sm * mean};
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool PW, bool STAG, int BKT>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm2_kernel(IGemmParams p, float* partial,
                                                                      int k_tiles_per_split) {
    // BKT = 32 with a 4-deep ring (prefetch distance 1.5 slabs of 64 at the same LDS footprint) was
    // built and measured: correct, but 10-20 % slower than BKT = 64 with 2 stages (a barrier per
    // 20 MFMAs costs more than the extra look-ahead buys); the shipped variants all use 64.
    static_assert(BKT == 64 || BKT == 32, "K slab depth");
    static_assert((BM / (512 / BKT)) % (WAVES_M * WAVES_N) == 0, "A-tile DMA rows must divide over the waves");
    static_assert((BM / WAVES_M) % 16 == 0 && (BN / WAVES_N) % 16 == 0, "wave tile must be MFMA-shaped");
    static_assert(STAGES >= 2 && STAGES <= 8, "ring depth");

    // The body is device-only: clang's host pass cannot type-check the gfx950 LDS-DMA builtin
    // (16-byte size) and would silently drop the kernel's host stub.
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOB = 0x80000000u;   // byte offset beyond any buffer -> DMA writes zeros
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NT = 64 * NW;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    // DMA instructions per wave per slab.  The B tile may not divide evenly over the waves
    // (160 rows / 8 waves): the first B_REM waves then issue one instruction more.
    // One DMA wave-instruction moves 1 KiB = RPI tile rows of BKT halves (8 rows x 128 B or 16 x 64 B).
    constexpr int CPR = BKT / 8;                // 16-byte chunks per tile row
    constexpr int RPI = 64 / CPR;               // tile rows per DMA instruction
    constexpr int KSTEPS = BKT / 32;            // MFMA k-steps per slab
    constexpr int A_PW = BM / RPI / NW;
    constexpr int B_INSTR = BN / RPI, B_REM = B_INSTR % NW, B_PW = (B_INSTR + NW - 1) / NW;
    constexpr int LPW = A_PW + B_PW;            // waves < B_REM (or all, when even)
    constexpr int LPW_LO = A_PW + B_PW - 1;     // waves >= B_REM when uneven
    constexpr int STAGE_HALVES = (BM + BN) * BKT;
    constexpr int LDC = BN + 8;
    using L = IGemm2Lds<BM, BN, STAGES, BKT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* ring = reinterpret_cast<half_t*>(smem);
    half_t* sC = reinterpret_cast<half_t*>(smem);
    float* sRed = reinterpret_cast<float*>(smem + L::EPI);      // [BM][BN / 8][2] row-statistics partials (producer)
    float* sG = reinterpret_cast<float*>(smem + L::EPI + L::RED);   // [NT][4] GroupNorm partials (producer)
    float* sStat = reinterpret_cast<float*>(smem + L::MAIN);    // [BM][2] mean, rstd of the block's rows (LN consumer)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

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
    const int split = blockIdx.y;
    const int nk_total = p.K / BKT;
    const int kt_begin = split * k_tiles_per_split;
    int nk = nk_total - kt_begin;
    if (nk > k_tiles_per_split) nk = k_tiles_per_split;

    // ---- DMA descriptors (wave-uniform) ----
    const long x_bytes = (long)p.N * p.H * p.W * p.ldx * 2;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, (int)x_bytes, 0x00020000);
    const long wrows = ((long)p.Cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, (int)(wrows * p.K * 2), 0x00020000);

    // ---- per-lane source coordinates: lane -> (row = RPI*instr + lane/CPR, chunk = (lane%CPR) ^ swz(row)):
    //      the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free, applied on the
    //      SOURCE side (128-byte rows: chunk ^ (row & 7); 64-byte rows: chunk ^ ((row >> 1) & 3)) ----
    auto swz = [](int r) { return BKT == 64 ? (r & 7) : ((r >> 1) & 3); };
    const int lrow = lane / CPR;
    const int chunk = (lane % CPR) ^ swz(lrow);
    const int OHW = p.OH * p.OW;
    const unsigned IH = (unsigned)(p.H << p.up), IW = (unsigned)(p.W << p.up);
    int a_ih[A_PW], a_iw[A_PW];   // top-left input coordinate of the row's pixel; rows >= M get a
                                  // coordinate that can never pass the unsigned bounds check
    unsigned a_base[A_PW];        // byte offset of (n, 0, 0, chunk), or of the row itself (pointwise)
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int m = m0 + (wave * A_PW + j) * RPI + lrow;
        const bool okm = m < p.M;
        const int mm = okm ? m : 0;
        if (PW) {
            a_ih[j] = okm ? 0 : -(1 << 28); a_iw[j] = 0;
            a_base[j] = (unsigned)(((long)mm * p.ldx + chunk * 8) * 2);
        } else {
            const int n = mm / OHW;
            const int rem = mm - n * OHW;
            const int oh = rem / p.OW;
            const int ow = rem - oh * p.OW;
            a_ih[j] = okm ? oh * p.stride - p.pad : -(1 << 28);
            a_iw[j] = ow * p.stride - p.pad;
            a_base[j] = (unsigned)(((long)n * p.H * p.W * p.ldx + chunk * 8) * 2);
        }
    }
    const bool b_hi = (B_REM == 0) || wave < B_REM;               // wave-uniform
    const int b_cnt = b_hi ? B_PW : B_PW - 1;
    const int b_first = b_hi ? wave * B_PW : B_REM * B_PW + (wave - B_REM) * (B_PW - 1);
    // GEGLU with a 128-column tile over two wave columns: the packed weight rows of a tile are
    // [64 hidden | 64 gate] (WeightStore::pack_geglu); the LDS image is filled as
    // [hidden 0-31 | gate 0-31 | hidden 32-63 | gate 32-63], so each wave's 64 columns hold 32 hidden units
    // next to their own gates and hidden * gelu(gate) is formed straight from the accumulators: no LDS
    // staging pass, no second barrier pair (that pass plus its reads was ~30 % of the GEGLU GEMM's time).
    constexpr bool GEGLU_REG = BN == 128 && WAVES_N == 2;
    const bool greg = GEGLU_REG && p.geglu;
    auto gperm = [](int r) { return r < 32 || r >= 96 ? r : (r < 64 ? r + 32 : r - 32); };   // LDS row -> packed row
    unsigned b_off[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int r = (b_first + j) * RPI + lrow;
        b_off[j] = (unsigned)((((long)(n0 + (greg ? gperm(r) : r))) * p.K + chunk * 8) * 2);
    }

    // tap state of the NEXT slab to issue
    // K order = [Cin/64][KH][KW][64] (misc.hip pack_conv_kernel): taps innermost
    int k0 = kt_begin * BKT;
    const int taps = p.KS * p.KS;
    constexpr int PER64 = 64 / BKT;             // slabs per 64-channel group (1 or 2)
    const int g64 = kt_begin / PER64;
    int sub = (kt_begin % PER64) * BKT;         // channel offset inside the 64-channel group
    int ci0 = (g64 / taps) * 64;
    const int tap0 = g64 % taps;
    int kh = tap0 / p.KS, kw = tap0 - kh * p.KS;
    const unsigned row_bytes = (unsigned)(p.ldx * 2);

    auto issue = [&](int slot) {
        half_t* sa = ring + slot * STAGE_HALVES;
        half_t* sb = sa + BM * BKT;
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            unsigned voff;
            if (PW) {
                voff = a_ih[j] >= 0 ? a_base[j] + (unsigned)(k0 * 2) : kOOB;
            } else {
                const int ih = a_ih[j] + kh, iw = a_iw[j] + kw;
                const bool ok = ((unsigned)ih < IH) & ((unsigned)iw < IW);   // negative -> huge unsigned
                const unsigned off = a_base[j] + (unsigned)((ih >> p.up) * p.W + (iw >> p.up)) * row_bytes +
                                     (unsigned)((ci0 + sub) * 2);
                voff = ok ? off : kOOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rx, (__attribute__((address_space(3))) void*)(sa + (wave * A_PW + j) * 512), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PW; ++j)
            if (j < b_cnt)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rw, (__attribute__((address_space(3))) void*)(sb + (b_first + j) * 512), 16,
                    b_off[j] + (unsigned)(k0 * 2), 0, 0, 0);
        k0 += BKT;
        sub += BKT;
        if (sub == 64) {
            sub = 0;
            if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ci0 += 64; } }
        }
    };

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    constexpr int D = STAGES - 1;       // prefetch distance
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nk) issue(s);

    // LayerNorm consumer: mean / rstd of the block's rows from the producer's partial sums, computed
    // while the first slabs are in flight (the compiler's vmcnt(0) ahead of these plain loads' first
    // use also covers the prologue DMAs, which the first step waits for anyway) and parked in LDS
    // beyond everything the main loop touches; the epilogue's first barrier publishes them.
    if (p.ln_stat) {
        static_assert(NT >= BM, "one thread per tile row");
        if (tid < BM) {
            const int m = m0 + tid;
            // all of the row's partials (at most kMaxLnParts: C <= 1280 behind 64-column producer tiles) are
            // loaded in one batch -- a loop with a runtime trip count issued them one L2 round trip at a time
            float2 pv[kMaxLnParts];
            const float2* src = reinterpret_cast<const float2*>(p.ln_stat) + (long)(m < p.M ? m : 0) * p.ln_parts;
            // (every load unconditional, at a clamped index: a select around a load makes hipcc branch and
            // wait per element)
#pragma unroll
            for (int k = 0; k < kMaxLnParts; ++k) pv[k] = src[k < p.ln_parts ? k : p.ln_parts - 1];
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxLnParts; ++k) {
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

    const int fr = lane & 15, fq = lane >> 4;
    // The K loop is unrolled by the ring depth, so a step's ring slot (and the slot its DMA refills) is
    // compile-time; a step's scalar work is the wait, the barrier and the issue's address updates.
    for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
        for (int slot = 0; slot < STAGES; ++slot) {
            const int kt = kt0 + slot;
            if (kt >= nk) break;
            // stage kt must have landed; at most D-1 younger stages may stay in flight
            {
                const int rem = nk - 1 - kt;
                const int y = rem < D - 1 ? rem : D - 1;
                if (b_hi) wait_stages<LPW, D - 1>(y); else wait_stages<LPW_LO, D - 1>(y);
            }
            __builtin_amdgcn_s_barrier();
            // (kt + D) % STAGES == (kt - 1) % STAGES.  With STAG the second wave group (the SIMD partners
            // of waves 0-3) issues its DMA after its MFMAs instead of before them, so on every SIMD one
            // wave is in its address/DMA-issue phase while the other feeds the matrix pipe.
            const bool late = STAG && wave >= NW / 2;
            if (!late && kt + D < nk) issue(slot == 0 ? STAGES - 1 : slot - 1);
            const half_t* cA = ring + slot * STAGE_HALVES;
            const half_t* cB = cA + BM * BKT;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                h8 fa[TM], fb[TN];
                const int ch = ks * 4 + fq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = wm * WTM + i * 16 + fr;
                    fa[i] = *reinterpret_cast<const h8*>(cA + r * BKT + ((ch ^ swz(r)) << 3));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = wn * WTN + j * 16 + fr;
                    fb[j] = *reinterpret_cast<const h8*>(cB + r * BKT + ((ch ^ swz(r)) << 3));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            if (late && kt + D < nk) issue(slot == 0 ? STAGES - 1 : slot - 1);
        }
    }

    // ---- split-K: raw fp32 partials, reduced by splitk_epilogue_kernel ----
    if (partial) {
        float* dst = partial + (long)split * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + fq * 4;
                if (m < p.M && n < p.Cout) *reinterpret_cast<f4*>(dst + (long)m * p.Cout + n) = acc[i][j];
            }
        }
        return;
    }

    // ---- fused epilogue through LDS (same as igemm.hip).  Every global load of the epilogue is issued
    //      in a batch ahead of its first use: one load-and-wait per accumulator tile (bias, row add) and
    //      per output chunk (residual) serialises 20-30 L2 round trips per block otherwise. ----
    if constexpr (GEGLU_REG) {
        if (p.geglu) {
            static_assert(!GEGLU_REG || TN == 4, "wave tile = 32 hidden + 32 gate columns");
            if (p.ln_stat) __syncthreads();          // the prologue's row statistics, written by other waves
            const int out_n0 = n0 >> 1;
            f4 bh[2], bg[2], wh[2], wg[2];
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                const int ch = n0 + gperm(wn * 64 + jh * 16 + fq * 4), cg = n0 + gperm(wn * 64 + 32 + jh * 16 + fq * 4);
                bh[jh] = *reinterpret_cast<const f4*>(p.bias + ch);
                bg[jh] = *reinterpret_cast<const f4*>(p.bias + cg);
                if (p.ln_stat) {
                    wh[jh] = *reinterpret_cast<const f4*>(p.ln_wsum + ch);
                    wg[jh] = *reinterpret_cast<const f4*>(p.ln_wsum + cg);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pr = wm * WTM + i * 16 + fr;
                const int m = m0 + pr;
                float mean = 0.f, rstd = 1.f;
                if (p.ln_stat) { mean = sStat[pr * 2]; rstd = sStat[pr * 2 + 1]; }
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    f4 hv = acc[i][jh], gv = acc[i][jh + 2];
                    if (p.ln_stat) { hv = (hv - mean * wh[jh]) * rstd; gv = (gv - mean * wg[jh]) * rstd; }
                    hv += bh[jh]; gv += bg[jh];
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)(hv[e] * gelu_erf_f(gv[e]));
                    if (m < p.M) *reinterpret_cast<h4*>(p.y + (long)m * p.ldy + out_n0 + wn * 32 + jh * 16 + fq * 4) = o;
                }
            }
            return;
        }
    }
    // (Measured and dropped for the plain GEMMs: storing straight from the accumulators as 8-byte pieces, as
    // the GEGLU path above does, instead of staging through LDS -- within noise on q|k|v / to_q shapes.)
    __syncthreads();     // every wave is done with the ring before it is overlaid
    f4 bias4[TN], wsum4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        bias4[j] = p.bias ? *reinterpret_cast<const f4*>(p.bias + n0 + wn * WTN + j * 16 + fq * 4) * p.bias_scale : f4{0.f, 0.f, 0.f, 0.f};
    if (p.ln_stat) {
#pragma unroll
        for (int j = 0; j < TN; ++j) wsum4[j] = *reinterpret_cast<const f4*>(p.ln_wsum + n0 + wn * WTN + j * 16 + fq * 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pr = wm * WTM + i * 16 + fr;
        const int m = m0 + pr;
        if (p.ln_stat) {     // y = rstd * (x W'^T - mean * wsum) + b'
            const float mean = sStat[pr * 2], rstd = sStat[pr * 2 + 1];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (acc[i][j] - mean * wsum4[j]) * rstd;
        }
        f4 add[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) add[j] = bias4[j];
        if (p.rowadd) {
            const float* ra = p.rowadd + (long)((m < p.M ? m : 0) / OHW) * p.rowadd_ld + n0 + wn * WTN + fq * 4;
            f4 r4[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                r4[j] = n0 + wn * WTN + j * 16 + fq * 4 < p.Cout ? *reinterpret_cast<const f4*>(ra + j * 16) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TN; ++j) add[j] += r4[j];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * WTN + j * 16 + fq * 4;
            f4 v = acc[i][j] * p.acc_scale + add[j];
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
            } else if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
            }
            h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<h4*>(sC + pr * LDC + col) = o;
        }
    }
    __syncthreads();

    // Output pass: thread t owns the 16-byte column chunk t % CH of the rows t / CH + k * RPP (the last
    // NT % CH threads sit out): a fixed column per thread, so the per-group sums a following GroupNorm
    // needs accumulate in registers across the thread's rows.
    constexpr int CH = BN / 8;
    constexpr int RPP = NT / CH;                 // tile rows per pass
    constexpr int ITER = (BM + RPP - 1) / RPP;
    const int c8 = tid % CH, rr = tid / CH;
    const int c = c8 * 8, n = n0 + c;
    const bool okc = rr < RPP && n < p.Cout;
    h8 rv[ITER];
    if (p.res) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int r = rr + it * RPP;
            const int m = m0 + r;
            rv[it] = (okc && r < BM && m < p.M) ? *reinterpret_cast<const h8*>(p.res + (long)m * p.ldres + n)
                                                : h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // GroupNorm partials: the 8 channels of the chunk fall into at most two groups (cpg >= 8 or cpg == 4)
    const int cpg = p.gnstat_out ? p.Cout / p.gn_groups : 8;
    const int g_first = n / cpg;
    const int gsplit = (g_first + 1) * cpg - n;          // channels e < gsplit belong to g_first
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
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
            if (p.gnstat_out) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e];
                    if (e < gsplit) { gs0 += f; gq0 += f * f; } else { gs1 += f; gq1 += f * f; }
                }
            }
        }
    }
    // Row statistics of the stored tile for a LayerNorm that follows (IGemmParams::rowstat_out): the
    // chunk partials are added per row in chunk order (fixed order: bitwise reproducible).
    if (p.rowstat_out) {
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            const int nch = (p.Cout - n0 < BN ? p.Cout - n0 : BN) / 8;
            float sm = 0.f, sq = 0.f;
            for (int k = 0; k < nch; ++k) { const float2 v = *reinterpret_cast<const float2*>(sRed + (tid * CH + k) * 2); sm += v.x; sq += v.y; }
            *reinterpret_cast<float2*>(p.rowstat_out + ((long)(m0 + tid) * p.rowstat_parts + tn) * 2) = float2{sm, sq};
        }
    }
    if (p.gnstat_out)
        gn_tile_stats<BM, BN, NT>(sG, tid, gs0, gq0, gs1, gq1, n0, p.Cout, cpg, p.gn_groups,
                                  p.gnstat_out + ((long)(m0 / OHW) * (OHW / BM) + (m0 % OHW) / BM) * p.gn_groups * 2);
#endif  // __HIP_DEVICE_COMPILE__
}

// Tile order for the launch (IGemmParams::mfast): weights bigger than the input activations -> keep a
// weight panel on one XCD.  SD_IGEMM_MFAST=0/1 overrides (tuning experiments).
inline int weights_outweigh_activations(const IGemmParams& p) {
    static const char* env = getenv("SD_IGEMM_MFAST");
    if (env) return atoi(env);
    const double wbytes = (double)p.Cout * p.K * 2.0;
    const double abytes = (double)p.N * p.H * p.W * p.Cin * 2.0;
    return wbytes > abytes ? 1 : 0;
}

// Fixed-order reduction of split-K partial slabs + the fused epilogue (bias, rowadd, residual).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(IGemmParams p, const float* __restrict__ partial,
                                                              int splits) {
    const long total = (long)p.M * (p.Cout / 8);
    const long slab = (long)p.M * p.Cout;
    const int OHW = p.OH * p.OW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / (p.Cout / 8);
        const int n = (int)(idx - m * (p.Cout / 8)) * 8;
        float v[8];
        const float* src = partial + m * p.Cout + n;
        const f4 a0 = *reinterpret_cast<const f4*>(src), a1 = *reinterpret_cast<const f4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[e + 4] = a1[e]; }
        for (int s = 1; s < splits; ++s) {
            const f4 b0 = *reinterpret_cast<const f4*>(src + s * slab), b1 = *reinterpret_cast<const f4*>(src + s * slab + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[e + 4] += b1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.acc_scale;
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += p.bias[n + e] * p.bias_scale;
        }
        if (p.rowadd) {
            const float* ra = p.rowadd + (m / OHW) * p.rowadd_ld + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += ra[e];
        }
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
        if (p.res) {
            const h8 rv = *reinterpret_cast<const h8*>(p.res + m * p.ldres + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rv[e]);
        }
        *reinterpret_cast<h8*>(p.y + m * p.ldy + n) = o;
    }
}

// The same reduction for an output a GroupNorm reads next, leaving that GroupNorm's summaries (IGemmParams::gnstat_out):
// one block = RB consecutive pixels of one image x a block of CB columns (whole groups, whole 16-byte chunks); every
// item (pixel, chunk) hands the sums / sums of squares of its stored fp16 values to LDS, split over the (at most two)
// groups the chunk touches, and one thread per group adds them in a fixed order: (mean, M2) of [RB pixels x the
// group's channels], the GnStats layout with rows = RB.  Split-K launches are the small maps (8 x 8 ... 32 x 32):
// without this their GroupNorm re-read the tensor for its own statistics.  RB = 4 / 16 by map size (at most 64
// summaries per image), CB <= 320: 512 blocks on the 8 x 8 level's 512 x 1280 outputs.
__global__ __launch_bounds__(256) void splitk_epilogue_gs_kernel(IGemmParams p, const float* __restrict__ partial, int splits,
                                                                 int RB, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sI = reinterpret_cast<float*>(smem);                // [RB][CH][4]
    const int c0 = blockIdx.y * CB;
    const int cw = p.Cout - c0 < CB ? p.Cout - c0 : CB;
    const int CH = cw / 8;
    const long slab = (long)p.M * p.Cout;
    const int OHW = p.OH * p.OW;
    const long m0 = (long)blockIdx.x * RB;
    const int cpg = p.Cout / p.gn_groups;
    for (int idx = threadIdx.x; idx < RB * CH; idx += 256) {
        const int r = idx / CH, c8 = idx - r * CH;
        const long m = m0 + r;
        const int n = c0 + c8 * 8;
        float v[8];
        const float* src = partial + m * p.Cout + n;
        const f4 a0 = *reinterpret_cast<const f4*>(src), a1 = *reinterpret_cast<const f4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[e + 4] = a1[e]; }
        for (int s = 1; s < splits; ++s) {
            const f4 b0 = *reinterpret_cast<const f4*>(src + s * slab), b1 = *reinterpret_cast<const f4*>(src + s * slab + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[e + 4] += b1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.acc_scale;
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += p.bias[n + e] * p.bias_scale;
        }
        if (p.rowadd) {
            const float* ra = p.rowadd + (m / OHW) * p.rowadd_ld + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += ra[e];
        }
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
        if (p.res) {
            const h8 rv = *reinterpret_cast<const h8*>(p.res + m * p.ldres + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rv[e]);
        }
        *reinterpret_cast<h8*>(p.y + m * p.ldy + n) = o;
        const int gsplit = (n / cpg + 1) * cpg - n;          // channels e < gsplit belong to the chunk's first group
        float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)o[e];
            if (e < gsplit) { gs0 += f; gq0 += f * f; } else { gs1 += f; gq1 += f * f; }
        }
        *reinterpret_cast<f4*>(sI + (long)idx * 4) = f4{gs0, gq0, gs1, gq1};
    }
    __syncthreads();
    const int gl = threadIdx.x;                              // group inside the column block
    if (gl < cw / cpg) {
        const int cfirst = (gl * cpg) / 8, clast = ((gl + 1) * cpg - 1) / 8;
        float sm = 0.f, sq = 0.f;
        for (int c8 = cfirst; c8 <= clast; ++c8) {
            const int off = ((c8 * 8) / cpg == gl) ? 0 : 2;
            for (int r = 0; r < RB; ++r) {
                const float2 v = *reinterpret_cast<const float2*>(sI + ((long)r * CH + c8) * 4 + off);
                sm += v.x; sq += v.y;
            }
        }
        const float mean = sm / (float)(RB * cpg);
        const long img = m0 / OHW, sl = (m0 - img * OHW) / RB;
        *reinterpret_cast<float2*>(p.gnstat_out + ((img * (OHW / RB) + sl) * p.gn_groups + c0 / cpg + gl) * 2) = float2{mean, sq - sm * mean};
    }
}
// Rows per block of that kernel for a problem (0: it cannot leave the summaries) and its column block
inline int splitk_gs_rows(const IGemmParams& p, int groups) {
    const int OHW = p.OH * p.OW;
    if (groups <= 0 || groups > 256 || p.Cout % groups != 0) return 0;
    const int cpg = p.Cout / groups;
    if (!(cpg >= 8 || cpg == 4)) return 0;
    for (int rb = 4; rb <= 16; rb <<= 1)
        if (OHW % rb == 0 && OHW / rb <= 64) return rb;
    return 0;
}
inline int splitk_gs_cols(const IGemmParams& p) {
    const int cpg = p.Cout / p.gn_groups;
    int unit = cpg;
    while (unit % 8 != 0) unit += cpg;                       // lcm(channels per group, 8)
    int cb = unit;
    while (cb + unit <= 320 && cb + unit <= p.Cout) cb += unit;
    return cb;
}
inline int launch_splitk_epilogue(const IGemmParams& p, const float* partial, int splits, hipStream_t s) {
    const int rb = p.gnstat_out ? splitk_gs_rows(p, p.gn_groups) : 0;
    if (rb) {
        const int cb = splitk_gs_cols(p);
        const size_t lds = (size_t)rb * (cb / 8) * 16;
        hipLaunchKernelGGL(splitk_epilogue_gs_kernel, dim3(p.M / rb, cdiv(p.Cout, cb)), dim3(256), lds, s, p, partial, splits, rb, cb);
    } else {
        const long total = (long)p.M * (p.Cout / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, p, partial, splits);
    }
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with the A operand kept in LDS across the nine taps.
//
// igemm2_kernel gathers a fresh [BM x 64] A tile per (64-channel slab, tap): nine gathers of the same
// rows, shifted by one pixel / one image row -- 9x the L2->LDS traffic and 9x the DMA instructions of
// what the slab needs, and FETCH_SIZE shows those re-reads leaking past the 4 MB L2 (218 MB per
// launch against 52 MB algorithmic on the 64x64-latent convs).  Here a block's 256 output pixels are
// an R x Wt patch of one image; per slab ONE halo tile [(R + 2) x (Wt + 2) pixels x 64 channels] is
// DMA'd (image borders read out of range and land as zeros), double-buffered, and the nine taps read
// it at shifted pixel offsets; only the weights [160 x 64 per tap] still stream, through a 3-deep
// ring.  The XOR swizzle is on the halo pixel index, so the 16 consecutive pixels of an MFMA operand
// stay conflict-free at every shift.  The 256 pixels are an R x Wt patch (64 / 32 / 16 columns wide,
// whichever tiles the image), so any H x W that such a patch divides qualifies; 8 waves (4 x 2).
// ---------------------------------------------------------------------------------------------
// Width of the 256-pixel patch a block covers: the widest of 64 / 32 / 16 columns that tiles the image
// (W % Wt == 0 and H % (256 / Wt) == 0); 0 if none does.
__host__ __device__ inline int halo_patch_width(int H, int W) {
    for (int wt = 64; wt >= 16; wt >>= 1)
        if (W % wt == 0 && H % (256 / wt) == 0) return wt;
    return 0;
}

// No GroupNorm summaries from this kernel's epilogue (igemm2_kernel has them): it sits on the 256-VGPR
// ceiling, and with the summary code in -- in any form tried: accumulated in the store loop, in a second
// loop, behind a noinline call -- the register allocator moves a dozen loop-invariant address registers to
// scratch and reloads them inside the main loop: 58.8 -> 72.7 us on the 64x64 320->320 conv, more than the
// 19 us statistics pass it would save.  (Its 8 spilled dwords stay outside the loop; check
// `scratch_` against the v_mfma range in the ISA after any change here.)
//
// GNF: GroupNorm (+ SiLU) of the INPUT fused in (diffusers ResnetBlock2D: norm -> SiLU -> conv, under
// sd_unified_pipeline.py:475-482 / :523).  The halo tile of a slab is transformed IN LDS, in place, after its DMA has
// landed and before its first tap reads it: y = silu(x * scale_c + shift_c) with scale_c = rstd_{n,g(c)} gamma_c,
// shift_c = beta_c - mean_{n,g(c)} scale_c.  Every thread transforms exactly the 16-byte slots it issued itself (so it
// knows in-image from border: border slots stay zero, the convolution pads the NORMALISED tensor), one slot per tap of
// the PREVIOUS slab's taps 3..8, i.e. next to that tap's 40 MFMAs; mean / rstd of the block's image are merged from the
// producer's (mean, M2) summaries in the prologue, the slab's 64 gamma | beta values arrive by one more DMA piece and
// are turned into scale | shift by wave 0 at tap 2.  The normalised tensor never exists in HBM (a 42 MB round trip and
// a launch per GroupNorm at the 64 x 64 level).
// KHU: the three kernel rows unrolled (all nine taps straight-line).  Unrolled, the 72 LDS addresses of the taps are
// hoisted out of the slab loop: 4-13 % faster, but the 160-column form then sits on the 256-VGPR ceiling (8 dwords
// spilled outside the loop) and has no room for the GroupNorm summaries in its epilogue; with the rows as a loop it
// needs ~190 registers and leaves them (62 us with summaries against 55 us without and 69 us for the best streaming tile
// that can, 64 x 64 level 320 -> 320).  So: unrolled unless the launch has to leave summaries at 160 columns.
template <int BN, bool GNF, bool KHU>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(IGemmParams p, float* partial, int slabs_per_split) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(BN == 160 || BN == 128, "column tile");
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int BM = 256, NW = 8, NT = 512, WAVES_N = 2;
    constexpr int WTM = 64, WTN = BN / 2, TM = 4, TN = WTN / 16;
    constexpr int B_INSTR = BN / 8, B_PW = (B_INSTR + NW - 1) / NW, B_REM = B_INSTR % NW;   // 160: 20 = 4 waves x 3 + 4 x 2; 128: 2 each
    constexpr int B_STAGE_HALVES = BN * 64;     // three ring stages: one per tap of a kernel row
    constexpr int AJ = 7;                        // halo DMA instructions per wave, at most
    constexpr int KH_UNROLL = KHU ? 3 : 1;
    constexpr int LDC = BN + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // tile = an R x Wt patch of one image (Wt = the widest of 64 / 32 / 16 that divides W, R = 256 / Wt)
    // (with the nearest-2x upsample fused in, the patch tiles the OUTPUT image and a halo pixel (ih, iw) of it reads
    // input pixel (ih >> 1, iw >> 1): the upsampled tensor never exists)
    const int W = p.OW, HW = p.OH * p.OW;                    // output image
    const int IW = p.W, IHW = p.H * p.W;                     // input image
    const int Wt = halo_patch_width(p.OH, W), Wp = Wt + 2;
    const int R = BM / Wt;
    const int HP = (R + 2) * Wp;                 // halo pixels per slab
    const int A_HALVES = HP * 64;
    half_t* sA = reinterpret_cast<half_t*>(smem);            // [2][HP][64], swizzled on the pixel index
    half_t* sB = sA + 2 * A_HALVES;                          // [3][160][64]
    half_t* sC = reinterpret_cast<half_t*>(smem);            // epilogue staging overlays both
    float* sCoef = reinterpret_cast<float*>(sB + 3 * B_STAGE_HALVES);   // GNF: [64 scale | 64 shift] of the slab in preparation
    float* sGS = sCoef + 128;                                // GNF: [groups][mean, rstd] of the block's image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (p.mfast) {
        const int tiles_m = gridDim.x / tiles_n;
        tn = bid / tiles_m; tm = bid - tn * tiles_m;
    } else {
        tm = bid / tiles_n; tn = bid - tm * tiles_n;
    }
    const int n0 = tn * BN;
    const int split = blockIdx.y;
    const int patches_w = W / Wt, patches = (p.OH / R) * patches_w;
    const int img = tm / patches, pidx = tm - img * patches;
    const int row0 = (pidx / patches_w) * R, col0 = (pidx % patches_w) * Wt;
    // GEMM row (NHWC pixel index) of the tile's local pixel ml
    auto row_of = [&](int ml) { const int r = ml / Wt; return img * HW + (row0 + r) * W + col0 + (ml - r * Wt); };
    const int nslab = p.Cin / 64;
    const int s_begin = split * slabs_per_split;
    int s_end = s_begin + slabs_per_split;
    if (s_end > nslab) s_end = nslab;

    const long x_bytes = (long)p.N * p.H * p.W * p.ldx * 2;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, (int)x_bytes, 0x00020000);
    const long wrows = ((long)p.Cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, (int)(wrows * p.K * 2), 0x00020000);

    // ---- halo DMA slots: slot = 64 * instr + lane = 8 * halo pixel + chunk position ----
    const int NI = (HP * 8 + 63) >> 6;
    int a_off[AJ];          // byte offset of the source chunk at slab 0; -2: outside the image (zeros); -1: no slot
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int i = wave + NW * j;
        const int slot = i * 64 + lane;
        const int hp = slot >> 3, cpos = slot & 7;
        const int hr = hp / Wp, hc = hp - hr * Wp;
        const int ih = row0 - 1 + hr, iw = col0 - 1 + hc;
        const bool inb = ((unsigned)ih < (unsigned)p.OH) & ((unsigned)iw < (unsigned)W);
        const int chunk = cpos ^ (hp & 7);
        int off = -1;
        if (i < NI && slot < HP * 8)
            off = inb ? (int)((((long)img * IHW + (long)(ih >> p.up) * IW + (iw >> p.up)) * p.ldx + chunk * 8) * 2) : -2;
        a_off[j] = off;
    }
    auto issueA = [&](int bufi, int slab) {
        half_t* dst = sA + bufi * A_HALVES;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = wave + NW * j;
            if (i < NI && a_off[j] != -1) {
                const unsigned voff = a_off[j] >= 0 ? (unsigned)a_off[j] + (unsigned)(slab * 128) : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + i * 512), 16,
                                                         voff, 0, 0, 0);
            }
        }
    };
    // ---- GNF: the slab's [64 gamma | 64 beta] (NormW::gb, packed per 64 channels) -> sCoef by one DMA piece of wave 0;
    //      wave 0 then turns them into scale | shift in place; a slot is normalised through them ----
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GNF ? p.gni_gb : nullptr), 0,
                                                                  GNF ? p.Cin * 8 : 0, 0x00020000);
    auto issueCoef = [&](int slab) {
        if (wave == 0 && lane < 32)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)sCoef, 16,
                                                     (unsigned)(slab * 512 + lane * 16), 0, 0, 0);
    };
    const float gni_inv_cpg = GNF ? 1.0f / (float)(p.Cin / p.gni_groups) : 0.f;
    auto convertCoef = [&](int slab) {               // wave 0, after its own wait for the piece
        if (wave == 0) {
            const int g = (int)(((float)(slab * 64 + lane) + 0.5f) * gni_inv_cpg);
            const float gam = sCoef[lane], bet = sCoef[64 + lane];
            const float sc = sGS[g * 2 + 1] * gam;
            sCoef[lane] = sc;
            sCoef[64 + lane] = bet - sGS[g * 2] * sc;
        }
    };
    const int chunkA = (lane & 7) ^ (lane >> 3);    // the 8-channel chunk of every halo slot this thread owns
    auto gnSlot = [&](int bufi, int j) {
        const int i = wave + NW * j;
        if (i < NI && a_off[j] >= 0) {               // in-image slot (border slots stay zero; -1: no slot)
            half_t* ptr = sA + bufi * A_HALVES + i * 512 + lane * 8;
            const h8 xv = *reinterpret_cast<const h8*>(ptr);
            const float* cs = sCoef + chunkA * 8;
            const f4 c0 = *reinterpret_cast<const f4*>(cs), c1 = *reinterpret_cast<const f4*>(cs + 4);
            const f4 d0 = *reinterpret_cast<const f4*>(cs + 64), d1 = *reinterpret_cast<const f4*>(cs + 68);
            h8 yv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = (float)xv[e] * c0[e] + d0[e];
                float b = (float)xv[e + 4] * c1[e] + d1[e];
                if (p.gni_silu) { a = silu_f(a); b = silu_f(b); }
                yv[e] = (half_t)a;
                yv[e + 4] = (half_t)b;
            }
            *reinterpret_cast<h8*>(ptr) = yv;
        }
    };
    // ---- weight tile: 8 rows of 64 halves per DMA instruction, chunk ^ (row & 7) on the source side ----
    const int lrow = lane >> 3;
    const int chunkB = (lane & 7) ^ lrow;
    const bool b_hi = B_REM == 0 || wave < B_REM;                // 160: waves 0-3 issue 3, waves 4-7 issue 2
    const int b_cnt = b_hi ? B_PW : B_PW - 1;
    const int b_first = b_hi ? wave * B_PW : B_REM * B_PW + (wave - B_REM) * (B_PW - 1);
    unsigned b_off[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j)
        b_off[j] = (unsigned)((((long)(n0 + (b_first + j) * 8 + lrow)) * p.K + chunkB * 8) * 2);
    auto issueB = [&](int stage, int g) {       // K order [Cin/64][KH][KW][64]: step g = slab * 9 + tap
        half_t* dst = sB + stage * B_STAGE_HALVES;
#pragma unroll
        for (int j = 0; j < B_PW; ++j)
            if (j < b_cnt)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + (b_first + j) * 512),
                                                         16, b_off[j] + (unsigned)(g * 128), 0, 0, 0);
    };

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    int hp0[TM];                                 // halo pixel of this lane's output pixel at tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * WTM + i * 16 + fr;
        const int r = ml / Wt, c = ml - r * Wt;
        hp0[i] = r * Wp + c;
    }

    int g = s_begin * 9;
    const int g_end = s_end * 9;
    if (g < g_end) {
        if constexpr (GNF) issueCoef(s_begin);
        issueA(0, s_begin);
        issueB(0, g);
        if (g + 1 < g_end) issueB(1, g + 1);
    }
    if constexpr (GNF) {
        // mean / rstd of the block's image from the producer's (mean, M2) summaries (Chan merges in a fixed order):
        // 512 / G threads per group take interleaved summaries, thread g merges those.  Scratch: halo buffer 1.
        float* red = reinterpret_cast<float*>(sA + A_HALVES);
        const int G = p.gni_groups, parts = NT / G;
        const float cpgf = (float)(p.Cin / G);
        {
            const int gi = tid % G, pi = tid / G;
            if (pi < parts) {
                float nA = 0.f, mA = 0.f, qA = 0.f;
                const float* src = p.gni_part + ((long)img * p.gni_S * G + gi) * 2;
                for (int k = pi; k < p.gni_S; k += parts) {
                    long rows = (long)IHW - (long)k * p.gni_rows;
                    if (rows > p.gni_rows) rows = p.gni_rows;
                    const float nB = (float)rows * cpgf, mB = src[(long)k * G * 2], qB = src[(long)k * G * 2 + 1];
                    const float n = nA + nB, d = mB - mA, f = nB / n;
                    mA += d * f; qA += qB + d * d * nA * f; nA = n;
                }
                red[(pi * G + gi) * 3] = nA; red[(pi * G + gi) * 3 + 1] = mA; red[(pi * G + gi) * 3 + 2] = qA;
            }
        }
        __syncthreads();                         // (also waits for the prologue DMA pieces: LDS writes in flight)
        if (tid < G) {
            float nA = red[tid * 3], mA = red[tid * 3 + 1], qA = red[tid * 3 + 2];
            for (int k = 1; k < parts; ++k) {
                const float nB = red[(k * G + tid) * 3], mB = red[(k * G + tid) * 3 + 1], qB = red[(k * G + tid) * 3 + 2];
                if (nB > 0.f) {
                    const float n = nA + nB, d = mB - mA, f = nB / n;
                    mA += d * f; qA += qB + d * d * nA * f; nA = n;
                }
            }
            const float var = qA / ((float)IHW * cpgf);
            sGS[tid * 2] = mA;
            sGS[tid * 2 + 1] = rsqrtf((var < 0.f ? 0.f : var) + p.gni_eps);
        }
        __syncthreads();
        if (g < g_end) {
            convertCoef(s_begin);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < AJ; ++j) gnSlot(0, j);
        }
    }
    // Main loop: slab -> kernel row -> the three taps of the row, unrolled.  The weight ring has three
    // stages, so the stage of a tap is its kw: compile-time, like the waits' immediates (the smallest
    // share any wave has: 2 weight instructions, 5 halo instructions).
    int abuf = 0;
    bool a_next = false;                         // halo of slab + 1 was issued at this slab's first tap
    for (int slab = s_begin; slab < s_end; ++slab) {
        const bool lastslab = slab + 1 >= s_end;
        const half_t* cA = sA + abuf * A_HALVES;
#pragma unroll KH_UNROLL
        for (int kh = 0; kh < 3; ++kh) {
            const bool lastrow = lastslab && kh == 2;
            const int rowoff = kh * Wp;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw, ++g) {
                // B(g) must have landed (and at the first tap the slab's halo, which is older).  Younger loads
                // that may stay in flight: B(g + 1), and the next halo while it is younger than B(g) (taps 1, 2).
                if (lastrow && kw == 2) wait_vmcnt<0>();
                else if (a_next && kh == 0 && kw >= 1) wait_vmcnt<7>();
                else wait_vmcnt<2>();
                // (GNF: this wave's in-place writes -- halo slots, scale | shift -- are done before it meets the barrier)
                if constexpr (GNF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (!(lastrow && kw >= 1)) issueB((kw + 2) % 3, g + 2);
                if (kw == 0 && kh == 0) {
                    a_next = !lastslab;
                    if (a_next) {
                        if constexpr (GNF) issueCoef(slab + 1);       // older than the halo pieces: landed by tap 2's wait
                        issueA(abuf ^ 1, slab + 1);
                    }
                }
                const half_t* cB = sB + kw * B_STAGE_HALVES;
                const int tapoff = rowoff + kw;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 fa[TM], fb[TN];
                    const int ch = ks * 4 + fq;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int hp = hp0[i] + tapoff;
                        fa[i] = *reinterpret_cast<const h8*>(cA + hp * 64 + ((ch ^ (hp & 7)) << 3));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int r = wn * WTN + j * 16 + fr;
                        fb[j] = *reinterpret_cast<const h8*>(cB + r * 64 + ((ch ^ (r & 7)) << 3));
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
                if constexpr (GNF) {
                    // The NEXT slab's halo is normalised during this slab's taps 3..8, one slot per tap (two in the last),
                    // after the tap's MFMAs: its DMA pieces have landed by tap 3's wait, scale | shift were written by
                    // wave 0 at tap 2 and published by tap 3's barrier.
                    // Measured (profiles/r03_gn_fused_conv.txt): the ~100 vector instructions per slot do NOT hide behind
                    // the MFMAs -- every wave of the block runs them at the same point of the tap, the matrix pipe idles
                    // meanwhile (+25-30 % per launch); running them first in one wave of each SIMD pair (+50 %) and a
                    // branch-free form for the scheduler to interleave (spills at full unroll) were worse.
                    if (a_next) {
                        const int T = kh * 3 + kw;
                        if (T == 2) convertCoef(slab + 1);
                        if (T >= 3) gnSlot(abuf ^ 1, T - 3);
                        if (T == 8) gnSlot(abuf ^ 1, 6);
                    }
                }
            }
        }
        abuf ^= 1;
    }

    // ---- split-K: raw fp32 partials, reduced by splitk_epilogue_kernel ----
    if (partial) {
        float* dst = partial + (long)split * p.M * p.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = row_of(wm * WTM + i * 16 + fr);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + fq * 4;
                if (n < p.Cout) *reinterpret_cast<f4*>(dst + (long)m * p.Cout + n) = acc[i][j];
            }
        }
        return;
    }

    // ---- fused epilogue through LDS (as igemm2_kernel: loads batched ahead of their use) ----
    __syncthreads();
    // (the two scales are read from the kernel arguments HERE, through volatile loads: as ordinary uses of `p` they are
    // fetched at kernel entry and held in scalar registers across the main loop, which the unrolled 160-column form paid
    // for with 16 more spilled registers, some reloaded inside the loop -- tools/check_spills.py)
    const float acc_scale = *reinterpret_cast<const volatile float*>(&p.acc_scale);
    const float bias_scale = *reinterpret_cast<const volatile float*>(&p.bias_scale);
    {
        f4 add[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
            add[j] = p.bias ? *reinterpret_cast<const f4*>(p.bias + n0 + wn * WTN + j * 16 + fq * 4) * bias_scale : f4{0.f, 0.f, 0.f, 0.f};
        if (p.rowadd) {          // one image per tile: the row add is the same for every row
            const float* ra = p.rowadd + (long)img * p.rowadd_ld + n0 + wn * WTN + fq * 4;
            f4 r4[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                r4[j] = n0 + wn * WTN + j * 16 + fq * 4 < p.Cout ? *reinterpret_cast<const f4*>(ra + j * 16) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TN; ++j) add[j] += r4[j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pr = wm * WTM + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * WTN + j * 16 + fq * 4;
                const f4 v = acc[i][j] * acc_scale + add[j];
                h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<h4*>(sC + pr * LDC + col) = o;
            }
        }
    }
    __syncthreads();
    // output pass with a fixed column chunk per thread (see igemm2_kernel): 25 rows x 20 chunks per pass
    constexpr int CH = BN / 8;
    constexpr int RPP = NT / CH;
    constexpr int ITER = (BM + RPP - 1) / RPP;
    const int c8 = tid % CH, rr = tid / CH;
    const int c = c8 * 8, n = n0 + c;
    const bool okc = rr < RPP && n < p.Cout;
    h8 rv[ITER];
    if (p.res) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int r = rr + it * RPP;
            rv[it] = (okc && r < BM) ? *reinterpret_cast<const h8*>(p.res + (long)row_of(r) * p.ldres + n)
                                     : h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // GroupNorm summaries of the stored tile (as igemm2_kernel's epilogue) -- in the 128-column form only: with 64
    // accumulator registers instead of 80 the allocator keeps the main loop free of spill reloads (the header
    // comment has what happened at 160 columns)
    constexpr bool GN = BN == 128 || !KHU;
    const int cpg = (GN && p.gnstat_out) ? p.Cout / p.gn_groups : 8;
    const int g_first = n / cpg;
    const int gsplit = (g_first + 1) * cpg - n;
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int r = rr + it * RPP;
        if (okc && r < BM) {
            h8 v = *reinterpret_cast<const h8*>(sC + r * LDC + c);
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[it][e]);
            }
            *reinterpret_cast<h8*>(p.y + (long)row_of(r) * p.ldy + n) = v;
            if (GN && p.gnstat_out) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e];
                    if (e < gsplit) { gs0 += f; gq0 += f * f; } else { gs1 += f; gq1 += f * f; }
                }
            }
        }
    }
    if constexpr (GN) {
        if (p.gnstat_out) {
            float* sG = reinterpret_cast<float*>(smem + ((BM * LDC * 2 + 15) & ~15));     // behind the staging tile
            gn_tile_stats<BM, BN, NT>(sG, tid, gs0, gq0, gs1, gq1, n0, p.Cout, cpg, p.gn_groups,
                                      p.gnstat_out + ((long)img * patches + pidx) * p.gn_groups * 2);
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

bool halo_supported(const IGemmParams& p) {
    return p.KS == 3 && p.stride == 1 && (p.up == 0 || p.up == 1) && p.pad == 1 && !p.geglu && !p.act && p.Cin % 64 == 0 &&
           p.Cout % 8 == 0 && halo_patch_width(p.OH, p.OW) > 0 && p.OH == p.H << p.up && p.OW == p.W << p.up &&
           p.K == 9 * p.Cin;
}

template <int BN, bool GNF, bool KHU>
int launch_halo_t(const IGemmParams& p, float* partial, int splits, hipStream_t s) {
    const int Wt = halo_patch_width(p.OH, p.OW), R = 256 / Wt;
    // [2 halo tiles | 3 weight stages | GNF: 128 floats scale / shift + 64 floats mean / rstd]
    size_t lds = (size_t)2 * (R + 2) * (Wt + 2) * 128 + (size_t)3 * BN * 128 + (GNF ? 768 : 0);
    const size_t epi = (size_t)256 * (BN + 8) * 2 + 16 + 512 * 16;       // staging tile + GroupNorm partials
    if (lds < epi) lds = epi;
    if (lds > 160 * 1024) { set_error("conv3x3_halo_kernel: LDS budget"); return 1; }
    static size_t attr_by_dev[64] = {};
    static std::mutex attr_mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    size_t& attr = attr_by_dev[dev & 63];
    if (lds > attr) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BN, GNF, KHU>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    const int tiles = (p.M / 256) * cdiv(p.Cout, BN);
    const int nslab = p.Cin / 64;
    const int per = cdiv(nslab, splits);
    const int eff_splits = cdiv(nslab, per);
    IGemmParams q = p;
    q.mfast = weights_outweigh_activations(p);
    if (eff_splits > 1) q.gnstat_out = nullptr;
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, GNF, KHU>), dim3(tiles, eff_splits), dim3(512), lds, s, q, eff_splits > 1 ? partial : nullptr,
                       per);
    SD_HIP_CHECK(hipGetLastError());
    if (eff_splits > 1) return launch_splitk_epilogue(p, partial, eff_splits, s);
    return 0;
}

template <int BN>
int launch_halo(const IGemmParams& p, float* partial, int splits, hipStream_t s) {
    if (p.gni_part) {
        if (p.gni_groups < 1 || p.gni_groups > 32 || 512 % p.gni_groups != 0 || p.Cin % p.gni_groups != 0 || !p.gni_gb) {
            set_error("conv3x3_halo_kernel: fused GroupNorm needs 1..32 groups dividing 512 and the packed affine");
            return 1;
        }
        return launch_halo_t<BN, true, false>(p, partial, splits, s);
    }
    if constexpr (BN == 160) {
        if (p.gnstat_out && splits <= 1) return launch_halo_t<BN, false, false>(p, partial, splits, s);
    }
    return launch_halo_t<BN, false, true>(p, partial, splits, s);
}

template <int BM, int BN, int WM, int WN, int STAGES, bool PW, bool STAG, int BKT>
int launch_v2p(const IGemmParams& p, float* partial, int splits, hipStream_t s) {
    constexpr size_t lds = IGemm2Lds<BM, BN, STAGES, BKT>::TOTAL;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BM, BN, WM, WN, STAGES, PW, STAG, BKT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int tiles = cdiv(p.M, BM) * cdiv(p.Cout, BN);
    const int nk = p.K / BKT;
    const int per = cdiv(nk, splits);
    const int eff_splits = cdiv(nk, per);
    IGemmParams q = p;
    q.mfast = weights_outweigh_activations(p);
    q.rowstat_parts = cdiv(p.Cout, BN);
    if (p.geglu && !(BN == 128 && WN == 2)) { set_error("igemm2: GEGLU needs a 128-column tile over two wave columns"); return 1; }
    if (eff_splits > 1 && (p.rowstat_out || p.ln_stat)) { set_error("igemm2: row statistics / LayerNorm fold need splits == 1"); return 1; }
    hipLaunchKernelGGL((igemm2_kernel<BM, BN, WM, WN, STAGES, PW, STAG, BKT>), dim3(tiles, eff_splits), dim3(64 * WM * WN), lds, s,
                       q, eff_splits > 1 ? partial : nullptr, per);
    SD_HIP_CHECK(hipGetLastError());
    if (eff_splits > 1) return launch_splitk_epilogue(p, partial, eff_splits, s);
    return 0;
}

template <int BM, int BN, int WM, int WN, int STAGES, bool STAG = false, int BKT = 64>
int launch_v2(const IGemmParams& p, float* partial, int splits, hipStream_t s) {
    if (p.KS == 1 && p.stride == 1 && p.up == 0)
        return launch_v2p<BM, BN, WM, WN, STAGES, true, STAG, BKT>(p, partial, splits, s);
    return launch_v2p<BM, BN, WM, WN, STAGES, false, STAG, BKT>(p, partial, splits, s);
}

// process-wide tuner / test override (sd_igemm_force): atomics, so that two handles driven from two threads see a
// consistent value each; the tile choice itself is per launch
std::atomic<int> g_force_variant{-1};
std::atomic<int> g_force_splits{0};

}  // namespace

// Variant ids (also used by the tuner in tests/tools): keep in sync with kIgemm2Names.
//   0: 256x128 8 waves 3 stages   1: 128x128 4 waves 2 stages   2: 128x160 4 waves 2 stages
//   3: 128x64 4 waves 2 stages    4: 64x64 4 waves 2 stages     5: 256x160 4 waves (4x1) 3 stages
//   6 / 7: 256x128 / 256x160, 8 waves, staggered DMA issue
//   10 / 15: conv3x3_halo_kernel<160> / <128> (256x160 / 256x128, A halo tile resident across the taps; 3x3 stride-1
//          convs, W in 16/32/64; the 128-column form for the VAE's 128 / 256 / 512 widths, and it can leave
//          GroupNorm summaries)
//   11 / 12: 128x80 (4 x 1 waves, 2- / 3-deep ring): M = 2048, N = 1280 is exactly 256 such tiles -- one per CU at the
//          least L2 -> LDS traffic a 256-tile grid can have there (133 MB against the 64x64 tile's 205 MB)
//   13 / 14: wsgemm.hip -- persistent, weight-stationary 128x160 / 128x128 (GEGLU) tiles for the K = 320 pointwise
//          problems of the 64x64 level; taken whenever wsgemm_supported() (SD_NO_WSGEMM=1 turns that off)
//   8 / 9: 128x64 / 128x160 with a 3-deep ring: only pays on the small-M, deep-K shapes of the 8x8 and
//          16x16 levels when their weights come cold from HBM (as they do inside a forward); with the
//          weights cache-resident the 2-deep rings win everywhere (tools/tune_igemm.py, SD_BENCH_COLD_MB)
// Measured and dropped in round 2 (profiles/tune/r02_deep_rings.txt): 4- and 6-deep rings on 128x64 / 64x64 /
// 128x128 / 128x160 tiles (one block per CU, 3-5 slabs in flight) for the small-grid linears of the 16x16 /
// 8x8 levels: 0-60 % slower than the 2- / 3-deep rings at 2-3 blocks per CU on every UNet shape.  Those
// launches are bound by the L2 -> LDS transfer rate per CU and by their prologue + epilogue, not by the
// per-slab round trip a deeper ring hides.  (The ring-depth template parameter still goes to 8.)
// printf formats of the kernel names as rocprofv3 prints them (%s = the pointwise flag)
static const char* kIgemm2Names[] = {
    "igemm2_kernel<256,128,4,2,3,%s,false,64>", "igemm2_kernel<128,128,2,2,2,%s,false,64>",
    "igemm2_kernel<128,160,2,2,2,%s,false,64>", "igemm2_kernel<128,64,2,2,2,%s,false,64>",
    "igemm2_kernel<64,64,2,2,2,%s,false,64>",   "igemm2_kernel<256,160,4,1,3,%s,false,64>",
    "igemm2_kernel<256,128,4,2,3,%s,true,64>",  "igemm2_kernel<256,160,4,2,3,%s,true,64>",
    "igemm2_kernel<128,64,2,2,3,%s,false,64>",  "igemm2_kernel<128,160,2,2,3,%s,false,64>",
    "conv3x3_halo_kernel<160>",
    "igemm2_kernel<128,80,4,1,2,%s,false,64>", "igemm2_kernel<128,80,4,1,3,%s,false,64>",
    "wsgemm_kernel<160,false,...>", "wsgemm_kernel<128,true,...>", "conv3x3_halo_kernel<128>",
    "igemm2_kernel<128,80,4,1,4,%s,false,64>", "igemm2_kernel<128,80,4,1,5,%s,false,64>", "igemm3_kernel"};
constexpr int kNumVariants = 19;

void igemm2_force(int variant, int splits) { g_force_variant = variant; g_force_splits = splits; }

bool igemm2_supported(const IGemmParams& p) {
    const long x_bytes = (long)p.N * p.H * p.W * p.ldx * 2;
    const long wrows = ((long)p.Cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad;
    return p.K % BK == 0 && p.Cin % BK == 0 && (p.Cout % 8) == 0 && x_bytes < (1L << 31) &&
           wrows * p.K * 2 < (1L << 31) && (!p.geglu || p.Cout % 128 == 0) && !(p.geglu && p.act);
}

static void tile_dims(int v, int* bm, int* bn) {
    static const int dims[kNumVariants][2] = {{256, 128}, {128, 128}, {128, 160}, {128, 64}, {64, 64}, {256, 160}, {256, 128}, {256, 160},
                                               {128, 64}, {128, 160}, {256, 160}, {128, 80}, {128, 80}, {128, 80}, {128, 128}, {256, 128},
                                               {128, 80}, {128, 80}, {128, 80}};
    *bm = dims[v][0]; *bn = dims[v][1];
}

// Per-shape choices measured on MI355X by tools/tune_igemm.py (profiles/tune/*.json).
struct TunedEntry { int M, N, K, ks, stride, up, geglu, variant, splits; float us; int alt_variant; float alt_us; };
static const TunedEntry kTuned[] = {
#include "igemm2_table.inc"
};

// Tile variant + split-K for a problem: the tuned table when the shape is in it, otherwise a rule
// distilled from the same measurements: 128x160 tiles whenever Cout is a multiple of 160 (every
// UNet / VAE width is), 128x128 otherwise; narrow tiles for small, shallow problems; split-K until
// about two blocks per CU (512) are in flight, keeping >= 12 K-slabs per slice.
static void igemm2_pick_raw(const IGemmParams& p, int* variant, int* splits);

static bool wsgemm_enabled() {
    static const bool off = getenv("SD_NO_WSGEMM") != nullptr;
    return !off;
}

bool igemm2_scales_ok(const IGemmParams& p) {
    if (!igemm2_supported(p) || p.geglu) return false;
    int v, sp;
    igemm2_pick(p, &v, &sp);
    return v != 13 && v != 14;
}

bool igemm2_gn_fusable(const IGemmParams& p, int groups) {
    return igemm2_supported(p) && halo_supported(p) && p.up == 0 && groups >= 1 && groups <= 32 && 512 % groups == 0 &&
           p.Cin % groups == 0;
}

void igemm2_pick(const IGemmParams& p, int* variant, int* splits) {
    igemm2_pick_raw(p, variant, splits);
    if (p.gni_part && *variant != 10 && *variant != 15) {
        // a GroupNorm is applied to the input inside the kernel: only the halo kernel does that.  160 columns where they
        // divide the width, split over the slabs until about one block per CU is in flight (>= 4 slabs per slice)
        *variant = p.Cout % 160 == 0 ? 10 : 15;
        const long tiles = (long)(p.M / 256) * cdiv(p.Cout, *variant == 10 ? 160 : 128);
        int sp = 1;
        const int nslab = p.Cin / 64;
        while (tiles * sp < 192 && nslab / (sp + 1) >= 4 && sp < 8) ++sp;
        *splits = sp;
    }
    if (p.ln_stat) *splits = 1;          // the LayerNorm correction lives in the fused epilogue only
    const bool ws_ok = wsgemm_supported(p);
    if ((*variant == 13 || *variant == 14) && !ws_ok) { *variant = p.geglu ? 1 : 2; *splits = 1; }     // (forced on a problem it does not take)
    // (with a residual the streamed tiles win by 1-6 us per launch -- profiles/r02_wsgemm.txt -- so those stay on them
    // unless variant 13 is forced)
    if (g_force_variant.load() < 0 && ws_ok && !p.res && wsgemm_enabled()) { *variant = p.geglu ? 14 : 13; *splits = 1; }
    if (*variant == 13 && p.geglu) *variant = 14;
    if (*variant == 14 && !p.geglu) *variant = 13;
}

bool igemm2_emits_rowstats(const IGemmParams& p, int* parts) {
    if (!igemm2_supported(p) || p.geglu) return false;
    int v, sp;
    IGemmParams q = p;                                  // the choice as it will be made WITH the statistics requested
    static float sentinel;
    if (!q.rowstat_out) q.rowstat_out = &sentinel;
    igemm2_pick(q, &v, &sp);
    if (sp > 1 || v == 10 || v == 15) return false;
    if (v == 13) { *parts = wsgemm_rowstat_parts(p); return *parts <= kMaxLnParts; }
    int bm, bn;
    tile_dims(v, &bm, &bn);
    *parts = cdiv(p.Cout, bn);
    return *parts <= kMaxLnParts;
}

bool igemm2_emits_gnstats(const IGemmParams& p, int groups, int* rows) {
    if (!igemm2_supported(p) || p.geglu || groups <= 0 || p.Cout % groups != 0) return false;
    const int cpg = p.Cout / groups;
    if (!(cpg >= 8 || cpg == 4)) return false;          // a 16-byte chunk may touch at most two groups
    int v, sp;
    IGemmParams q = p;                                  // the choice as it will be made WITH the summaries requested
    static float sentinel;
    q.gnstat_out = &sentinel;
    igemm2_pick(q, &v, &sp);
    if (v == 13 || v == 14) return false;
    if (sp > 1) {                                       // split-K: the reduction kernel leaves them
        const int rb = splitk_gs_rows(p, groups);
        if (!rb) return false;
        *rows = rb;
        return true;
    }
    if ((v == 10 || v == 15) && !halo_supported(p)) v = 7;
    int bm, bn;
    tile_dims(v, &bm, &bn);
    const int OHW = p.OH * p.OW;
    if (OHW % bm != 0) return false;                    // every tile inside one image
    if (bn % cpg != 0 && p.Cout > bn) return false;     // group boundaries on tile boundaries
    *rows = bm;
    return true;
}

static void igemm2_pick_raw(const IGemmParams& p, int* variant, int* splits) {
    if (g_force_variant.load() >= 0) {
        *variant = g_force_variant.load();
        *splits = g_force_splits.load() > 0 ? g_force_splits.load() : 1;
        if (p.geglu && *variant != 0 && *variant != 1 && *variant != 6 && *variant != 13 && *variant != 14) *variant = 1;
        if (p.geglu || p.act) *splits = 1;
        if ((*variant == 10 || *variant == 15) && !halo_supported(p)) *variant = 7;
        return;
    }
    for (const TunedEntry& e : kTuned)
        if (!p.act && e.M == p.M && e.N == p.Cout && e.K == p.K && e.ks == p.KS && e.stride == p.stride && e.up == p.up &&
            e.geglu == p.geglu) {
            *variant = e.variant; *splits = e.splits;
            if ((*variant == 10 || *variant == 15) && !halo_supported(p)) *variant = 7;   // (same M x N x K from another image shape)
            return;
        }
    const int nk = p.K / BK;
    int v;
    if (p.act) {                 // the activation lives in the fused epilogue only: never split K
        v = (long)cdiv(p.M, 128) * cdiv(p.Cout, 64) >= 256 ? 3 : 4;
        if ((long)cdiv(p.M, 128) * cdiv(p.Cout, 128) >= 256) v = 1;
        *variant = v;
        *splits = 1;
        return;
    }
    if (p.geglu) v = (long)cdiv(p.M, 256) * cdiv(p.Cout, 128) >= 512 ? 0 : 1;
    else if (p.Cout % 160 == 0) v = 2;
    else v = 1;
    int bm, bn;
    tile_dims(v, &bm, &bn);
    long tiles = (long)cdiv(p.M, bm) * cdiv(p.Cout, bn);
    if (!p.geglu && tiles < 256 && nk <= 40) {          // small and shallow: narrow tiles fill more CUs
        v = (long)cdiv(p.M, 128) * cdiv(p.Cout, 64) >= 256 ? 3 : 4;
        tile_dims(v, &bm, &bn);
        tiles = (long)cdiv(p.M, bm) * cdiv(p.Cout, bn);
    }
    int sp = 1;
    if (!p.geglu && tiles < 384 && nk >= 24) {
        sp = (int)((512 + tiles / 2) / tiles);
        if (sp > 8) sp = 8;
        while (sp > 1 && nk / sp < 12) --sp;
    }
    *variant = v;
    *splits = sp;
}

const char* igemm2_name(int variant) { return kIgemm2Names[variant]; }

long igemm2_partial_floats(const IGemmParams& p) {
    int v, sp;
    igemm2_pick(p, &v, &sp);
    return sp > 1 ? (long)sp * p.M * p.Cout : 0;
}

int launch_igemm2(const IGemmParams& p, float* partial, hipStream_t s) {
    int v, sp;
    igemm2_pick(p, &v, &sp);
    if (sp > 1 && !partial) sp = 1;
    static const bool pg_k320 = getenv("SD_PGEMM_K320") != nullptr;       // A/B: the K = 320 GEGLU on it instead of wsgemm
    if (p.geglu && ((v != 13 && v != 14) || pg_k320) && g_force_variant.load() < 0 && pgemm_geglu_supported(p))
        return launch_pgemm_geglu(p, s);
    switch (v) {
        case 0: return launch_v2<256, 128, 4, 2, 3>(p, partial, sp, s);
        case 1: return launch_v2<128, 128, 2, 2, 2>(p, partial, sp, s);
        case 2: return launch_v2<128, 160, 2, 2, 2>(p, partial, sp, s);
        case 3: return launch_v2<128, 64, 2, 2, 2>(p, partial, sp, s);
        case 4: return launch_v2<64, 64, 2, 2, 2>(p, partial, sp, s);
        case 5: return launch_v2<256, 160, 4, 1, 3>(p, partial, sp, s);
        case 6: return launch_v2<256, 128, 4, 2, 3, true>(p, partial, sp, s);
        case 7: return launch_v2<256, 160, 4, 2, 3, true>(p, partial, sp, s);
        case 8: return launch_v2<128, 64, 2, 2, 3>(p, partial, sp, s);
        case 9: return launch_v2<128, 160, 2, 2, 3>(p, partial, sp, s);
        case 10: return launch_halo<160>(p, partial, sp, s);
        case 15: return launch_halo<128>(p, partial, sp, s);
        case 11:
        case 12:
        case 18:
            // 18 = igemm3_kernel (activation operand through the ordinary load path instead of the LDS-DMA ring) asked for by
            // name (tests, tuner), falling back to the 3-stage DMA tile for problems it does not take
            {
                static const bool g3_auto = getenv("SD_IGEMM3") != nullptr;        // off by default: measured not faster (igemm3.hip)
                if (sp <= 1 && (v == 18 || (g3_auto && g_force_variant.load() < 0)) && igemm3_supported(p))
                    return launch_igemm3(p, weights_outweigh_activations(p), s);
            }
            if (v == 11) return launch_v2<128, 80, 4, 1, 2>(p, partial, sp, s);
            return launch_v2<128, 80, 4, 1, 3>(p, partial, sp, s);
        case 16: return launch_v2<128, 80, 4, 1, 4>(p, partial, sp, s);
        case 17: return launch_v2<128, 80, 4, 1, 5>(p, partial, sp, s);
        case 13:
        case 14: return launch_wsgemm(p, s);
        default: set_error("igemm2: bad variant"); return 1;
    }
}

}  // namespace sd
