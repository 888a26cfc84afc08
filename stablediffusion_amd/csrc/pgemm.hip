// Persistent GEGLU projection (gfx950 / CDNA4): y[m, j] = hidden_j * gelu(gate_j), [hidden | gate] = LN(x) W^T + b, the
// first linear of diffusers' FeedForward (GEGLU) under BasicTransformerBlock (call site
// /root/reference/pipelines/sd_unified_pipeline.py:475-482), for the 32 x 32 and 16 x 16 levels of the UNet
// (8192 x 5120 x 640, 2048 x 10240 x 1280: 1.2 ms of the 10 ms C2 forward at 590-630 TFLOP/s, 24 % MFMA busy).
//
// igemm2_kernel runs these as 640 / 1280 independent 256 x 128 tiles, five / two and a half per CU one after the other,
// each paying its own prologue (two slabs of L2 -> LDS latency with nothing to compute) and its LayerNorm-statistics
// round trip: s_memtime stamps put prologue + epilogue at 41 % of a tile.  Here a block is PERSISTENT: it owns one M tile
// (256 rows) and a run of consecutive N tiles, and its LDS-DMA ring never stops at a tile boundary -- the slabs of tile
// t + 1 are already in flight while tile t finishes and runs its epilogue.  What that needs:
//   * the A rows are the same for every tile of the run (re-streamed from L2 per tile, W streams once), so the LayerNorm
//     mean / rstd of the block's 256 rows are computed ONCE, in the prologue, before any DMA is in flight;
//   * the GEGLU epilogue works straight from the accumulators (no LDS staging: the ring keeps filling underneath it);
//   * counted waits: a slab's wait allows the D - 1 younger slabs AND, for the first D slabs behind an epilogue, that
//     epilogue's 8 stores (issued after those slabs' DMA pieces, so younger in vmcnt order).
// Same operand layout, swizzle, MFMA tiling (8 waves, 4 x 2, 64 x 64 per wave) and GEGLU weight packing as
// igemm2_kernel<256, 128, 4, 2, 3>; blocks that share an M tile get consecutive ids on one XCD (its A rows stay in that L2).
#include <cstdlib>

#include "kernels.h"

namespace sd {
namespace {

constexpr int PG_BM = 256, PG_BN = 128, PG_STAGES = 3, PG_NST = 8;
constexpr int kPgMaxLnParts = 20;

template <int N>
__device__ __forceinline__ void pg_wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct PgLds {
    static constexpr int RING = PG_STAGES * (PG_BM + PG_BN) * 64 * 2;
    static constexpr int TOTAL = RING + PG_BM * 8;
};

__global__ __launch_bounds__(512) void geglu_persist_kernel(IGemmParams p, int tiles_n, int chunks, int pg_stag) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int BM = PG_BM, BN = PG_BN, STAGES = PG_STAGES, D = STAGES - 1;
    constexpr int NW = 8, WAVES_N = 2, WTM = 64, WTN = 64, TM = 4, TN = 4;
    constexpr int A_PW = BM / 8 / NW, B_PW = BN / 8 / NW, LPW = A_PW + B_PW;        // DMA pieces per wave per slab: 4 + 2
    constexpr int STAGE_HALVES = (BM + BN) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* ring = reinterpret_cast<half_t*>(smem);
    float* sStat = reinterpret_cast<float*>(smem + PgLds::RING);      // [BM][2] mean, rstd of the block's rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / chunks, ch = bid - tm * chunks;
    const int tn_lo = (int)((long)ch * tiles_n / chunks), tn_hi = (int)((long)(ch + 1) * tiles_n / chunks);
    const int T = tn_hi - tn_lo;
    const int m0 = tm * BM;
    const int nk = p.K / 64;

    // ---- LayerNorm statistics of the block's rows (plain loads: nothing else is in flight yet) ----
    if (p.ln_stat) {
        if (tid < BM) {
            float2 pv[kPgMaxLnParts];
            const float2* src = reinterpret_cast<const float2*>(p.ln_stat) + (long)(m0 + tid) * p.ln_parts;
#pragma unroll
            for (int k = 0; k < kPgMaxLnParts; ++k) pv[k] = src[k < p.ln_parts ? k : p.ln_parts - 1];
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int k = 0; k < kPgMaxLnParts; ++k) {
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
    __syncthreads();
    float mean[TM], rstd[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pr = wm * WTM + i * 16 + fr;
        mean[i] = p.ln_stat ? sStat[pr * 2] : 0.f;
        rstd[i] = p.ln_stat ? sStat[pr * 2 + 1] : 1.f;
    }

    // ---- DMA descriptors and per-lane source coordinates (as igemm2_kernel, pointwise) ----
    const long x_bytes = (long)p.M * p.ldx * 2;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, (int)x_bytes, 0x00020000);
    const long wrows = ((long)p.Cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad;
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, (int)(wrows * p.K * 2), 0x00020000);
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (lrow & 7);
    unsigned a_base[A_PW], b_base[B_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const long m = m0 + (wave * A_PW + j) * 8 + lrow;
        a_base[j] = (unsigned)((m * p.ldx + chunk * 8) * 2);
    }
    // GEGLU: LDS rows [hidden 0-31 | gate 0-31 | hidden 32-63 | gate 32-63] of the packed [64 hidden | 64 gate]
    auto gperm = [](int r) { return r < 32 || r >= 96 ? r : (r < 64 ? r + 32 : r - 32); };
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int r = (wave * B_PW + j) * 8 + lrow;
        b_base[j] = (unsigned)((((long)tn_lo * BN + gperm(r)) * p.K + chunk * 8) * 2);
    }
    const unsigned b_tile_bytes = (unsigned)((long)BN * p.K * 2);

    // the stream: slab gi = t * nk + kt of the run; state of the NEXT slab to issue
    int i_kt = 0, i_slot = 0;
    unsigned i_boff = 0;                 // byte offset of the issue tile's W rows from the run's first tile
    int i_left = T;                      // tiles not yet fully issued
    auto issue = [&]() {
        half_t* sa = ring + i_slot * STAGE_HALVES;
        half_t* sb = sa + BM * 64;
        const bool live = i_left > 0;
#pragma unroll
        for (int j = 0; j < A_PW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sa + (wave * A_PW + j) * 512), 16,
                                                     live ? a_base[j] + (unsigned)(i_kt * 128) : kOOB, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < B_PW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sb + (wave * B_PW + j) * 512), 16,
                                                     live ? b_base[j] + i_boff + (unsigned)(i_kt * 128) : kOOB, 0, 0, 0);
        if (++i_kt == nk) { i_kt = 0; i_boff += b_tile_bytes; --i_left; }
        if (++i_slot == STAGES) i_slot = 0;
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue();         // (past the end of the run: out-of-range pieces, zeros, never read)

    int c_slot = 0;
    for (int t = 0; t < T; ++t) {
        f4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            // slab (t, kt) has landed: the D - 1 younger slabs stay in flight, and behind an epilogue its stores too
            // (every slab of the stream is issued -- the tail as out-of-range pieces -- so the count never shrinks)
            if (t > 0 && kt < D) pg_wait_vmcnt<(D - 1) * LPW + PG_NST>(); else pg_wait_vmcnt<(D - 1) * LPW>();
            __builtin_amdgcn_s_barrier();
            // slab D ahead, into the stage every wave finished reading one step ago.  The second wave group (the SIMD
            // partners of waves 0-3) issues after its MFMAs instead of before them, so that on every SIMD one wave is in
            // its DMA-issue stretch while the other feeds the matrix pipe (igemm2_kernel's STAG)
            const bool late = pg_stag && wave >= NW / 2;
            if (!late) issue();
            const half_t* cA = ring + c_slot * STAGE_HALVES;
            const half_t* cB = cA + BM * 64;
            if (++c_slot == STAGES) c_slot = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 fa[TM], fb[TN];
                const int chk = ks * 4 + fq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = wm * WTM + i * 16 + fr;
                    fa[i] = *reinterpret_cast<const h8*>(cA + r * 64 + ((chk ^ (r & 7)) << 3));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = wn * WTN + j * 16 + fr;
                    fb[j] = *reinterpret_cast<const h8*>(cB + r * 64 + ((chk ^ (r & 7)) << 3));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            if (late) issue();
        }
        // ---- epilogue of tile t, straight from the accumulators (the next tile's first slabs are landing meanwhile):
        //      a wave's 64 columns are 32 hidden units next to their own gates ----
        const int n0 = (tn_lo + t) * BN;
        const int out_n0 = n0 >> 1;
        f4 bh[2], bg[2], wh[2], wg[2];
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            const int chh = n0 + gperm(wn * 64 + jh * 16 + fq * 4), cg = n0 + gperm(wn * 64 + 32 + jh * 16 + fq * 4);
            bh[jh] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                         __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, (int)(wrows * 4), 0x00020000), chh * 4, 0, 0));
            bg[jh] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                         __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, (int)(wrows * 4), 0x00020000), cg * 4, 0, 0));
            if (p.ln_stat) {
                wh[jh] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                             __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_wsum), 0, (int)(wrows * 4), 0x00020000), chh * 4, 0, 0));
                wg[jh] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                             __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_wsum), 0, (int)(wrows * 4), 0x00020000), cg * 4, 0, 0));
            }
        }
        // (those loads are younger than everything in flight: the compiler's wait for them would be vmcnt(0), draining the
        // ring, if it saw the DMA pieces as loads it must respect; it counts them like any VMEM op -- checked in the ISA)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                f4 hv = acc[i][jh], gv = acc[i][jh + 2];
                if (p.ln_stat) { hv = (hv - mean[i] * wh[jh]) * rstd[i]; gv = (gv - mean[i] * wg[jh]) * rstd[i]; }
                hv += bh[jh]; gv += bg[jh];
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)(hv[e] * gelu_erf_f(gv[e]));
                *reinterpret_cast<h4*>(p.y + (long)m * p.ldy + out_n0 + wn * 32 + jh * 16 + fq * 4) = o;
            }
        }
    }
    pg_wait_vmcnt<0>();          // the tail's out-of-range pieces: let them land before the block leaves
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

bool pgemm_geglu_supported(const IGemmParams& p) {
    static const bool off = getenv("SD_NO_PGEMM") != nullptr;
    if (off || !p.geglu || !(p.KS == 1 && p.stride == 1 && p.up == 0)) return false;
    if (p.M % PG_BM != 0 || p.Cout % PG_BN != 0 || p.K % 64 != 0 || p.K < 128) return false;
    if (p.res || p.rowadd || p.act || p.rowstat_out || p.gnstat_out || !p.bias) return false;
    if (p.ln_stat && p.ln_parts > kPgMaxLnParts) return false;
    if ((long)p.M * p.ldx * 2 >= (1L << 31)) return false;
    const long tiles = (long)(p.M / PG_BM) * (p.Cout / PG_BN);
    return tiles >= 512 && p.M / PG_BM <= 256;                  // at least two tiles per block on average
}

int launch_pgemm_geglu(const IGemmParams& p, hipStream_t s) {
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&geglu_persist_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, PgLds::TOTAL));
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int tiles_m = p.M / PG_BM, tiles_n = p.Cout / PG_BN;
    // blocks = M tiles x chunks: one block per CU, the N tiles of an M tile split into `chunks` runs
    int chunks = cus / tiles_m;
    if (chunks < 1) chunks = 1;
    if (chunks > tiles_n) chunks = tiles_n;
    static const int stag = getenv("SD_PGEMM_STAG") ? atoi(getenv("SD_PGEMM_STAG")) : 0;      // (measured: no difference)
    hipLaunchKernelGGL(geglu_persist_kernel, dim3(tiles_m * chunks), dim3(512), PgLds::TOTAL, s, p, tiles_n, chunks, stag);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace sd
