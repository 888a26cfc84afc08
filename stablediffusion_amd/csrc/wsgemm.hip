// Weight-stationary pointwise GEMM for the shallow (K = 320) linears of the 64x64 level (gfx950 / CDNA4).
//
// igemm2_kernel streams both operands per output tile; for y[M, N] = x[M, 320] W[N, 320]^T with M = 32768
// that is 184 KB of L2 -> LDS traffic per 128x160 tile (the per-CU L2 -> LDS rate, ~70 GB/s, is what bounds
// those launches), a ~1 us pipeline fill per tile, and an epilogue that only overlaps another block's main
// loop by luck.  Here a block keeps one N tile of the weights -- all of K -- resident in LDS and streams M
// tiles of the activations past it:
//   * 256 persistent blocks, 8 waves (4 x 2), tile 128 x BN, wave tile 32 x BN/2; BN = 160 (plain) or 128
//     (GEGLU: 32 hidden + their 32 gate columns per wave, the packing of WeightStore::pack_geglu);
//   * LDS: W tile [K/64][BN][64] (100 KB / 80 KB), a ring of 3 / 4 activation slabs [128 x 64], the
//     LayerNorm partials of two M tiles, bias and wsum; everything arrives by LDS-DMA, swizzled on the source;
//   * the activation stream never stops at a tile boundary: slab s + D is issued at step s whatever tile it
//     belongs to, so there is no per-tile pipeline fill;
//   * the epilogue of tile t - 1 (LayerNorm correction, bias, GEGLU / residual, row statistics, stores) runs
//     in five chunks inside the five K steps of tile t, from a second accumulator set, its VALU instructions
//     placed between the MFMAs (sched_group_barrier); the W fragments of step s + 1 are read during the MFMAs
//     of step s (W never changes inside a run), so only the four A fragment reads sit behind a step's barrier;
//   * XCD x (blocks x, x + 8, ...) owns the M tiles [x, x + 1) * tiles_m / 8 for every N tile: the
//     activations of that range (2.6 MB at M = 32768) are fetched into ONE L2 and re-read from there by the
//     N tiles; a block takes a contiguous run of that XCD's (n tile, m tile) list and reloads W when the
//     n tile changes (at most twice).
// Waits for the LDS-DMA slabs are counted by hand: vmcnt(N) with N = the operations this wave issued after the
// slab it needs (loads, stores, atomics and LDS-DMA retire in issue order in vmcnt).  The residual is read with
// plain buffer loads, which hipcc counts itself (together with the DMA builtins and its stores).
//
// Measured (profiles/r02_wsgemm.txt, MI355X, 8 x 64 x 64 x 320 input, L2-warm): N = 960 34 us (igemm2 best 37),
// GEGLU N = 2560 87 us (100), N = 320 15 us (15); the whole UNet forward 10.49 ms against 10.56 ms.  A
// steady-state K step takes ~1500 cycles for 640 cycles of MFMA per SIMD: the s_memtime timeline and the
// ablations (DMA stream alone 23 us for GEGLU -- at the L2 -> LDS rate; MFMA + DMA 55 us; epilogue + DMA 46 us)
// say the step is bound by issue: LDS fragment reads (14 KB per wave per step, ~220 cycles to issue), the MFMAs
// and the epilogue's VALU share each SIMD's issue port, and the two waves of a SIMD get little overlap out of
// it.  A two-barrier ping-pong of the wave halves (one half in its MFMA segment while the other reads) was
// built on this kernel and measured the same to 5 % slower (34 / 92 us), as was running the halves' MFMA and
// epilogue parts in opposite order.
#include <type_traits>

#include <cstdlib>

#include "kernels.h"

namespace sd {
namespace {

typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void ws_wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ u2 ws_load_b64(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    return __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
}

constexpr int kWsK = 320, kWsNK = kWsK / 64;
constexpr int kWsStatParts = 4;                  // LayerNorm partials per row the consumer path takes, at most

template <int BN>
struct WsLds {
    static constexpr int SLOTS = BN == 160 ? 3 : 4;
    static constexpr int B_BYTES = kWsNK * BN * 128;
    static constexpr int A_BYTES = SLOTS * 128 * 128;
    static constexpr int STAT_BYTES = 2 * kWsStatParts * 1024;      // two tiles' worth
    static constexpr int TOTAL = B_BYTES + A_BYTES + STAT_BYTES + BN * 8;
};

template <int BN, bool GEGLU, bool LN, bool RES, bool RS>
__global__ __launch_bounds__(512) void wsgemm_kernel(IGemmParams p, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!GEGLU || (BN == 128 && !RES && !RS), "GEGLU: 128-column tile, no residual / row statistics");
    constexpr unsigned kOOB = 0x80000000u;       // byte offset beyond any buffer -> the DMA writes zeros
    constexpr int BM = 128, NK = kWsNK, NW = 8;
    constexpr int WTN = BN / 2, TM = 2, TN = WTN / 16;
    constexpr int SLOTS = WsLds<BN>::SLOTS, D = SLOTS - 1, R = 2;
    constexpr int NCH = GEGLU ? 4 : TN;          // epilogue chunks per tile, one per K step
    static_assert(NCH <= NK, "one epilogue chunk per K step");
    constexpr int SLOT_HALVES = BM * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sB = reinterpret_cast<half_t*>(smem);
    half_t* sA = reinterpret_cast<half_t*>(smem + WsLds<BN>::B_BYTES);
    float* sStat = reinterpret_cast<float*>(smem + WsLds<BN>::B_BYTES + WsLds<BN>::A_BYTES);
    float* sBias = sStat + WsLds<BN>::STAT_BYTES / 4;
    float* sWsum = sBias + BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (lrow & 7);   // XOR swizzle on the DMA source (128-byte rows)

    // ---- this block's share: XCD x = blockIdx % 8 owns a range of M tiles, its blocks split the
    //      (n tile, m tile) list of that range into contiguous runs ----
    const int xcd = blockIdx.x & 7, kblk = blockIdx.x >> 3, nblk = gridDim.x >> 3;
    const int mt_lo = (int)((long)xcd * tiles_m / 8), mt_hi = (int)((long)(xcd + 1) * tiles_m / 8);
    const int nm = mt_hi - mt_lo;
    const long L = (long)tiles_n * nm;
    int lo = (int)(kblk * L / nblk);
    const int hi = (int)((kblk + 1) * L / nblk);

    const long x_bytes = (long)p.M * p.ldx * 2;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, (int)x_bytes, 0x00020000);
    const long wrows = ((long)p.Cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, (int)(wrows * p.K * 2), 0x00020000);
    const int parts = LN ? p.ln_parts : 1;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(LN ? p.ln_stat : p.bias), 0, LN ? (int)((long)p.M * parts * 8) : 0, 0x00020000);

    // (p.dbg_unchecked, tests only: the residual's descriptor without its range check -- every residual load then has to
    // be in range by the index arithmetic alone, which tests/test_ops_gpu.py::test_wsgemm_residual_loads_need_no_range_check
    // verifies bit for bit; see profiles/r02_wsgemm.txt for the fault this settles)
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(RES ? p.res : p.x), 0, RES ? (p.dbg_unchecked ? 0x7ffffff0 : (int)((long)p.M * p.ldres * 2)) : 0, 0x00020000);

    // GEGLU: LDS rows [hidden 0-31 | gate 0-31 | hidden 32-63 | gate 32-63] of the packed [64 hidden | 64 gate]
    auto perm = [](int r) { return !GEGLU || r < 32 || r >= 96 ? r : (r < 64 ? r + 32 : r - 32); };
    const int ocol_w = GEGLU ? wn * 32 + fq * 4 : wn * WTN + fq * 4;     // this lane's column inside the output tile

    // per-lane LDS read offsets (halves): fragment row r, k-chunk (ks * 4 + fq) ^ (r & 7); r & 7 == fr & 7
    int offA[TM][2], offB[TN][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + fq) ^ (fr & 7)) << 3;
#pragma unroll
        for (int i = 0; i < TM; ++i) offA[i][ks] = (wm * 32 + i * 16 + fr) * 64 + c;
#pragma unroll
        for (int j = 0; j < TN; ++j) offB[j][ks] = (wn * WTN + j * 16 + fr) * 64 + c;
    }
    while (lo < hi) {
        const int tn = lo / nm, mo = lo - tn * nm;
        int T = nm - mo;
        if (T > hi - lo) T = hi - lo;
        const int mt0 = mt_lo + mo;
        lo += T;
        const int n0 = tn * BN;

        __syncthreads();                         // the previous run is done with every LDS region
        // ---- the stationary operand: W rows [n0, n0 + BN) x all of K, bias, wsum ----
        for (int ii = wave; ii < NK * (BN / 8); ii += NW) {
            const int kt = ii / (BN / 8), q = ii - kt * (BN / 8);
            const unsigned off = (unsigned)((((long)(n0 + perm(q * 8 + lrow))) * p.K + kt * 64 + chunk * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rw, (__attribute__((address_space(3))) void*)(sB + (kt * BN + q * 8) * 64), 16, off, 0, 0, 0);
        }
        if (tid < BN) {
            sBias[tid] = p.bias ? p.bias[n0 + perm(tid)] : 0.f;
            if (LN) sWsum[tid] = p.ln_wsum[n0 + perm(tid)];
        }

        // ---- activation stream state: slab gi = t * NK + kt of this run, issued D steps ahead ----
        int ti = 0, kti = 0, slot_i = 0;
        unsigned a_off[2];
        const unsigned a_tile_bytes = (unsigned)(BM * p.ldx * 2);           // x_bytes < 2^31: every tile offset fits
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long m = (long)mt0 * BM + (wave * 2 + j) * 8 + lrow;
            a_off[j] = (unsigned)((m * p.ldx + chunk * 8) * 2);
        }
        auto tile_a_base = [&](int t) {          // (M % 128 == 0: only the tiles past the run read out of range)
#pragma unroll
            for (int j = 0; j < 2; ++j) a_off[j] = t < T ? a_off[j] + a_tile_bytes : kOOB;
        };
        auto issue_a = [&]() {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rx, (__attribute__((address_space(3))) void*)(sA + slot_i * SLOT_HALVES + (wave * 2 + j) * 512), 16,
                    a_off[j] + (unsigned)(kti * 128), 0, 0, 0);
            if (++kti == NK) { kti = 0; ++ti; tile_a_base(ti); }
            if (++slot_i == SLOTS) slot_i = 0;
        };
#pragma unroll
        for (int s = 0; s < D; ++s) issue_a();

        // residual of (tile te, chunk c, row half i): 4 halves per lane, prefetched R steps ahead (plain buffer
        // loads: hipcc counts them -- and the LDS-DMA builtins and its stores -- in its own vmcnt waits)
        u2 resv[NCH][TM];
        auto res_off = [&](int te, int c, int i) {
            const long m = (long)(mt0 + te) * BM + wm * 32 + i * 16 + fr;
            return (unsigned)((m * p.ldres + n0 + ocol_w + c * 16) * 2);
        };
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i) resv[c][i] = u2{0u, 0u};
        ws_wait_vmcnt<0>();
        __syncthreads();

        f4 acc[TM][TN], prev[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) { acc[i][j] = f4{0.f, 0.f, 0.f, 0.f}; prev[i][j] = acc[i][j]; }
        float mean[TM] = {0.f, 0.f}, rstd[TM] = {1.f, 1.f};
        float rsum[TM] = {0.f, 0.f}, rsq[TM] = {0.f, 0.f};
        int slot_c = 0;
        h8 fb[TN][2];                                    // W fragments of the step about to run (prefetched by the step before)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j][ks] = *reinterpret_cast<const h8*>(sB + offB[j][ks]);

        // One tile position of the run: FIRST = tile 0's K steps only, STEADY = tile t's K steps with the epilogue of
        // tile t - 1 in them, LAST = the epilogue of the last tile on its own.  A K step: [wait, barrier] [A fragment
        // reads] [residual prefetch] [slab DMA] [statistics DMA] [MFMAs, with the epilogue chunk and the next step's W
        // fragment reads between them].  The slab wait's count N = the operations (residual loads, DMA, stores) of
        // the D - 1 whole steps since the slab it needs was issued, a compile-time function of (mode, kt, whether the
        // tile position before had an epilogue); stores are counted (left out, every wait would also ask for the last
        // stores to be acknowledged).  The slab's own step contributes only the statistics DMA; hipcc moving an
        // earlier operation of that step below the slab only makes the wait stricter.
        auto tile_steps = [&](auto modec, auto prevc, int t) {
            constexpr int MODE = decltype(modec)::value;     // 0 FIRST, 1 STEADY, 2 LAST
            constexpr bool PREV_EPI = decltype(prevc)::value;   // the tile position before this one ran an epilogue (stores)
            constexpr bool MMA = MODE != 2, EPI = MODE != 0;
            half_t* yrow[TM];
            long mrow[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                mrow[i] = (long)(mt0 + t - 1) * BM + wm * 32 + i * 16 + fr;
                yrow[i] = p.y + mrow[i] * p.ldy + (GEGLU ? n0 / 2 : n0) + ocol_w;
            }
            auto step = [&](auto ktc) {
                constexpr int KT = decltype(ktc)::value;
                // loads of step k of a tile in mode m (k < 0: the tile before, whose steps 3 and 4 look the same in
                // FIRST and STEADY): residual prefetch for the chunk R steps on, when a tile consumes it
                constexpr auto stat_at = [](int k) { k = ((k % NK) + NK) % NK; return (LN && k == 1) ? 1 : 0; };
                constexpr auto res_at = [](int m, int k) {
                    if (!RES) return 0;
                    if (k < 0) { k += NK; m = 1; }
                    if ((k + R) % NK >= NCH) return 0;
                    const bool next_tile = (k + R) / NK == 1;        // consumed by the epilogue of THIS tile (runs one tile on)
                    return (next_tile ? m != 2 : m != 0) ? 2 : 0;
                };
                // stores of step k: the epilogue chunk's (and the row statistics with the last chunk)
                constexpr auto stores_at = [](int m, int k) {
                    bool epi = m != 0;
                    if (k < 0) { k += NK; epi = PREV_EPI; }
                    if (!epi || k >= NCH) return 0;
                    return GEGLU ? 1 : TM + ((RS && k == NCH - 1) ? TM : 0);
                };
                constexpr auto ops_at = [=](int m, int k) { return res_at(m, k) + 2 + stat_at(k) + stores_at(m, k); };
                // the wait at the END of step KT, for the slab of step KT + 1 (issued in step KT + 1 - D): everything of the
                // D - 1 steps since (this one included) may stay in flight
                // the wait at the start of step KT, for the slab issued in step KT - D: the operations of the D - 1 whole
                // steps since may stay in flight (the slab's own step: only what the source issues after the slab)
                constexpr int NA = [=]() { int n = stat_at(KT - D); for (int u = 1; u < D; ++u) n += ops_at(MODE, KT - u); return n; }();

                h8 fa[TM][2];
                if constexpr (MMA) {
                    ws_wait_vmcnt<NA>();        // slab (t, KT) has landed; D - 1 younger slabs stay in flight
                    __builtin_amdgcn_s_barrier();
                    const half_t* cA = sA + slot_c * SLOT_HALVES;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i][ks] = *reinterpret_cast<const h8*>(cA + offA[i][ks]);
                    if (++slot_c == SLOTS) slot_c = 0;
                }
                if constexpr (res_at(MODE, KT) != 0) {
                    constexpr int cc = (KT + R) % NK;
#pragma unroll
                    for (int i = 0; i < TM; ++i) resv[cc][i] = ws_load_b64(rr, res_off(t - 1 + (KT + R) / NK, cc, i));
                }
                if constexpr (MMA) {
                    issue_a();
                    if constexpr (LN && KT == 1) {   // LayerNorm partials of tile t, read by its epilogue one tile on; two
                                                     // buffers, so the tile before can still be reading its own
                        const int seg = wave % parts;
                        const unsigned off = (unsigned)(((long)(mt0 + t) * BM * parts * 8) + seg * 1024 + lane * 16);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            rs, (__attribute__((address_space(3))) void*)(sStat + (t & 1) * (kWsStatParts * 256) + seg * 256), 16, off, 0, 0, 0);
                    }
                }
                auto epi_part = [&]() {
                    if constexpr (LN && KT == 0) {       // mean / rstd of the previous tile's rows from the producer's partials
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const float2* src = reinterpret_cast<const float2*>(sStat + ((t - 1) & 1) * (kWsStatParts * 256)) +
                                                (wm * 32 + i * 16 + fr) * parts;
                            float sm = 0.f, sq = 0.f;
#pragma unroll
                            for (int k = 0; k < kWsStatParts; ++k) {
                                const float2 v = src[k < parts ? k : parts - 1];
                                sm += k < parts ? v.x : 0.f;
                                sq += k < parts ? v.y : 0.f;
                            }
                            const float inv = 1.0f / (float)p.ln_C;
                            mean[i] = sm * inv;
                            float var = sq * inv - mean[i] * mean[i];
                            var = var < 0.f ? 0.f : var;
                            rstd[i] = rsqrtf(var + p.ln_eps);
                        }
                    }
                    if constexpr (KT < NCH) {
                        if constexpr (GEGLU) {
                            constexpr int i = KT >> 1, jh = KT & 1;
                            const int cl = wn * 64 + jh * 16 + fq * 4;           // LDS-image row of the hidden columns
                            f4 hv = prev[i][jh], gv = prev[i][jh + 2];
                            if (LN) {
                                hv = (hv - mean[i] * *reinterpret_cast<const f4*>(sWsum + cl)) * rstd[i];
                                gv = (gv - mean[i] * *reinterpret_cast<const f4*>(sWsum + cl + 32)) * rstd[i];
                            }
                            hv += *reinterpret_cast<const f4*>(sBias + cl);
                            gv += *reinterpret_cast<const f4*>(sBias + cl + 32);
                            h4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (half_t)(hv[e] * gelu_erf_f(gv[e]));
                            *reinterpret_cast<h4*>(yrow[i] + jh * 16) = o;
                        } else {
                            constexpr int j = KT;
                            const int cl = wn * WTN + j * 16 + fq * 4;
                            const f4 b4 = *reinterpret_cast<const f4*>(sBias + cl);
                            f4 w4 = f4{0.f, 0.f, 0.f, 0.f};
                            if (LN) w4 = *reinterpret_cast<const f4*>(sWsum + cl);
#pragma unroll
                            for (int i = 0; i < TM; ++i) {
                                f4 v = prev[i][j];
                                if (LN) v = (v - mean[i] * w4) * rstd[i];
                                v += b4;
                                h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                                if constexpr (RES) {
                                    const h4 rv = __builtin_bit_cast(h4, resv[j][i]);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)o[e] + (float)rv[e]);
                                }
                                *reinterpret_cast<h4*>(yrow[i] + j * 16) = o;
                                if constexpr (RS) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { const float f = (float)o[e]; rsum[i] += f; rsq[i] += f * f; }
                                }
                            }
                            if constexpr (RS && j == NCH - 1) {      // the wave's 80 columns of each row: one partial
#pragma unroll
                                for (int i = 0; i < TM; ++i) {
                                    float sm = rsum[i], sq = rsq[i];
                                    sm += __shfl_xor(sm, 16); sq += __shfl_xor(sq, 16);
                                    sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
                                    if (fq == 0)
                                        *reinterpret_cast<float2*>(p.rowstat_out + (mrow[i] * p.rowstat_parts + tn * 2 + wn) * 2) = float2{sm, sq};
                                    rsum[i] = 0.f; rsq[i] = 0.f;
                                }
                            }
                        }
                    }
                                };
                if constexpr (MMA) {
                    // this step's MFMAs on the A fragments just read and the W fragments the step before prefetched; between
                    // them the epilogue chunk of the tile before (VALU, stores) and the W fragment reads of the next step
                    // (W never changes inside a run, so those reads need no barrier)
                    const half_t* nB = sB + ((KT + 1) % NK) * BN * 64;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
                            fb[j][ks] = *reinterpret_cast<const h8*>(nB + offB[j][ks]);     // reloaded in place for the next step
                        }
                    if constexpr (EPI) epi_part();
                    constexpr int VPM = EPI ? (GEGLU ? 7 : 3) : 0;
#pragma unroll
                    for (int q = 0; q < 2 * TM * TN; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (VPM) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                        if (q < 2 * TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
                    epi_part();
                }
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});
            if constexpr (MMA) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) { prev[i][j] = acc[i][j]; acc[i][j] = f4{0.f, 0.f, 0.f, 0.f}; }
            }
        };
        tile_steps(std::integral_constant<int, 0>{}, std::false_type{}, 0);
        if (T > 1) tile_steps(std::integral_constant<int, 1>{}, std::false_type{}, 1);
        for (int t = 2; t < T; ++t) {
            tile_steps(std::integral_constant<int, 1>{}, std::true_type{}, t);
        }
        tile_steps(std::integral_constant<int, 2>{}, std::true_type{}, T);
        ws_wait_vmcnt<0>();          // slabs issued past the end of the run read out of range (zeros): let them land
    }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BN, bool GEGLU, bool LN, bool RES, bool RS>
int launch_ws(const IGemmParams& p, hipStream_t s) {
    constexpr size_t lds = WsLds<BN>::TOTAL;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wsgemm_kernel<BN, GEGLU, LN, RES, RS>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    IGemmParams q = p;
    q.rowstat_parts = 2 * (p.Cout / BN);
    if (RES && getenv("SD_WS_RES_UNCHECKED")) q.dbg_unchecked = 1;       // (test switch, read per launch)
    const int tiles_m = p.M / 128, tiles_n = p.Cout / BN;
    hipLaunchKernelGGL((wsgemm_kernel<BN, GEGLU, LN, RES, RS>), dim3(256), dim3(512), lds, s, q, tiles_m, tiles_n);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

bool wsgemm_supported(const IGemmParams& p) {
    if (!(p.KS == 1 && p.stride == 1 && p.up == 0) || p.K != kWsK || p.Cin != kWsK) return false;
    if (p.M % 128 != 0 || p.M / 128 < 8 || p.rowadd || p.act || p.gnstat_out) return false;
    if ((long)p.M * p.ldx * 2 >= (1L << 31)) return false;
    if (p.ln_stat && (p.ln_parts > kWsStatParts || p.ln_parts < 1)) return false;
    if (p.geglu) return p.Cout % 128 == 0 && !p.res && !p.rowstat_out;
    if (p.Cout % 160 != 0) return false;
    if (p.ln_stat && (p.res || p.rowstat_out)) return false;     // combinations the UNet does not issue
    return true;
}

int wsgemm_rowstat_parts(const IGemmParams& p) { return p.Cout / 80; }

int launch_wsgemm(const IGemmParams& p, hipStream_t s) {
    if (!wsgemm_supported(p)) { set_error("wsgemm: unsupported problem"); return 1; }
    if (p.geglu) return p.ln_stat ? launch_ws<128, true, true, false, false>(p, s) : launch_ws<128, true, false, false, false>(p, s);
    if (p.ln_stat) return launch_ws<160, false, true, false, false>(p, s);
    if (p.res) return p.rowstat_out ? launch_ws<160, false, false, true, true>(p, s) : launch_ws<160, false, false, true, false>(p, s);
    return p.rowstat_out ? launch_ws<160, false, false, false, true>(p, s) : launch_ws<160, false, false, false, false>(p, s);
}

}  // namespace sd
