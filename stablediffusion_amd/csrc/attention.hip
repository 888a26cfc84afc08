// Flash-style attention for gfx950 (CDNA4, wave64): out = softmax(q k^T / sqrt(d)) v, fp16 in/out,
// fp32 scores / running max / output accumulators; the T x T score matrix is never materialised.
// Replaces F.scaled_dot_product_attention under diffusers' AttnProcessor2_0 for the UNet self/cross
// attention and the VAE mid-block attention (call sites
// /root/reference/pipelines/sd_unified_pipeline.py:475-482, :523).
//
// MI355X-specific structure:
//   * "swapped" products so every softmax quantity of a query lives in one lane:
//       S^T = K Q^T   (MFMA A = K rows from LDS, B = Q fragments held in registers)
//       O^T = V^T P^T (MFMA A = V^T fetched with ds_read_b64_tr_b16 from a row-major V tile,
//                      B = P^T taken straight from the S^T accumulators -- no LDS round trip)
//   * small head dims (SD1.5 level 0: d = 40) are VALU-bound, not MFMA-bound, so the softmax is
//     stripped to ~4 VALU ops per score: max on the raw scores (v_max3), one FMA folding the
//     1/sqrt(d)*log2(e) scale and the running max, a raw v_exp_f32, and v_cvt_pkrtz_f16_f32 packing
//     two probabilities per instruction.  Where the head dim leaves spare rows in the last 16-row
//     PV tile (d = 40 -> 48) a column of ones is appended to V, so the softmax denominator falls
//     out of the PV MFMA for free; the truncation bias of pkrtz then cancels in the normalisation.
//   * the output rescale by exp2(m_old - m_new) is skipped (wave-uniformly) while the running max
//     grows by less than 2^8 -- probabilities stay below 256, well inside fp16.
//   * K / V tiles of 64 keys in LDS with row strides that are odd multiples of 32 bytes:
//     conflict-free for both ds_read_b128 fragment reads and the transposed reads.  Tiles are
//     double-buffered in LDS and filled by LDS-DMA (buffer_load ... lds, 16 B per lane) one tile
//     ahead: no staging registers, no VALU address work in the loop, ONE barrier per tile (PMC
//     showed waves 40 % of their time in s_waitcnt/barrier with two).  The padded LDS rows map to
//     DMA slots; pad chunks are masked off so the zero / ones padding written once survives, rows
//     past the last key read out of range and land as zeros.  Against staging through registers:
//     -8 % at d = 64, -11 % at d = 80, -18 % at d = 160 (d = 40 is bound by exp / cvt / max issue).
//     Measured and rejected: 128-key tiles (fewer barriers, more registers: 10-30 % slower) and
//     64 queries per wave for d = 64 / 80 (slower); d = 40 runs 64 queries per wave (QT = 4).
//   * head dims 40 / 80 / 160 (SD1.5), 64 (SDXL), 512 (VAE), 32 / 128 (test configs); the QK^T
//     contraction is zero-padded to a multiple of 32, the PV row tiles to a multiple of 16.
#include <cstdlib>

#include "kernels.h"

namespace sd {
namespace {

[[maybe_unused]] constexpr float kRescaleThreshold = 8.0f;   // log2 units

constexpr int odd32_bytes(int bytes) { return ((((bytes + 31) / 32) | 1)) * 32; }
// LDS row stride of the K tile in halves.  d = 40 rows are stored DENSE (80 B): 16 consecutive rows at a
// 20-dword stride start on 16 different bank quads, so the ds_read_b128 fragment reads stay conflict-free,
// and the tile is 5 DMA pieces instead of 10 (half of the padded image was masked-off pad slots; the DMA
// issue was 18 % of the kernel in a compile-time ablation).  The QK^T contraction still runs to 64: columns
// 40..63 of a row are the next row's first values (the V tile's, after the last row) times the ZERO pad of
// the query fragment.
constexpr int k_row_halves(int D) { return D == 40 ? 40 : odd32_bytes((D + 31) / 32 * 32 * 2) / 2; }

typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_rtz(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
}

// NWV: waves per block (4 or 8).  Eight waves share one K / V tile stream: half the DMA pieces and barrier episodes per
// query (the DMA issue was 18 % of the d = 40 kernel in the round-2 ablation), two waves per SIMD from ONE block.
template <int D, int QT, int KT, bool PRESC, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void attn_kernel(const half_t* __restrict__ q,
                                                   const half_t* __restrict__ k,
                                                   const half_t* __restrict__ v,
                                                   half_t* __restrict__ out, int Tq, int Tk_all,
                                                   int heads, long ldq, long ldk, long ldv,
                                                   long ldo, float scale_log2e, int causal, int qblocks) {
    static_assert(KT == 64 || KT == 128, "keys per tile");
    // The body is device-only: clang's host pass cannot type-check the gfx950 16-byte LDS-DMA builtin
    // inside a template and would silently drop the kernel's host stub (same as igemm2.hip).
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NSUB = KT / 16;                        // 16-key subtiles per tile
    constexpr int NKK = KT / 32;                         // 32-key k-steps of the PV product
    constexpr int DK = (D + 31) / 32 * 32;
    constexpr int KS = DK / 32;
    // PV row tiles; head dims that fill their last tile (32, 64) get one more, so that the ones column
    // can ride there: 4 extra MFMAs per key tile buy back 32 v_add_f32 per lane (VALU issue is the limit)
    constexpr int DT = (D + 15) / 16 + ((D % 16 == 0 && D <= 64) ? 1 : 0);
    constexpr int KSTR = k_row_halves(D);                // halves
    constexpr int VSTR = odd32_bytes(DT * 16 * 2) / 2;   // halves
    constexpr int CH = D / 8;                            // real 16-byte chunks per row
    constexpr int NCH = KT * CH;                         // chunk slots per tile (K and V alike)
    constexpr bool ONES = DT * 16 > D;                   // spare PV rows -> denominator via MFMA
    constexpr bool PREFETCH = D <= 160;                  // register-staged prefetch of the next tile
    constexpr int QB = 16 * QT * NWV;                    // queries per block
    constexpr int NTH = 64 * NWV;

    // With PREFETCH the K/V tiles are double-buffered in LDS (one barrier per tile); buffer b lives
    // at smem + b * TILE_HALVES.
    constexpr int TILE_HALVES = KT * (KSTR + VSTR);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sK = reinterpret_cast<half_t*>(smem);
    half_t* sV = sK + KT * KSTR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // 1-D grid, (batch, head) slow and query block fast, pushed through the XCD remap: the blocks that
    // run on one XCD are then a contiguous id range = whole heads, so a head's K / V tiles are pulled
    // into ONE 4 MB L2 instead of all eight (r1 PMC: 213 MB fetched per launch against 63 MB
    // algorithmic with the 2-D grid, whose 16 query blocks of a head were dealt round-robin over the XCDs)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = bid / qblocks, qblk = bid - bh * qblocks;
    const int b = bh / heads, h = bh - b * heads;
    const int q0 = qblk * QB + wave * (16 * QT);

    const half_t* qb = q + (long)b * Tq * ldq + h * D;
    // causal (CLIP text encoders): keys past the block's last query never contribute, so the key
    // loop simply ends there; inside it key j > query i is masked like a ragged tail
    int Tk = Tk_all;
    if (causal && qblk * QB + QB < Tk) Tk = qblk * QB + QB;
    const half_t* kb = k + (long)b * Tk_all * ldk + h * D;
    const half_t* vb = v + (long)b * Tk_all * ldv + h * D;

    // ---- one-time LDS padding: K columns [D, DK) = 0 (the Q pad is zero too), V columns
    //      [D, VSTR) = 0 with a column of ones at D when the denominator rides on the PV MFMA ----
    constexpr int KPAD = KSTR - D, VPAD = VSTR - D;
#pragma unroll
    for (int bufi = 0; bufi < (PREFETCH ? 2 : 1); ++bufi) {
        if constexpr (KPAD > 0) {
            for (int idx = tid; idx < KT * KPAD; idx += NTH) {
                const int r = idx / KPAD, c = D + idx - r * KPAD;
                sK[bufi * TILE_HALVES + r * KSTR + c] = (half_t)0.f;
            }
        }
        if constexpr (VPAD > 0) {
            for (int idx = tid; idx < KT * VPAD; idx += NTH) {
                const int r = idx / VPAD, c = D + idx - r * VPAD;
                sV[bufi * TILE_HALVES + r * VSTR + c] = (ONES && c == D) ? (half_t)1.f : (half_t)0.f;
            }
        }
    }

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[query fr][32 ks + 8 fq .. +8]
    h8 qf[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = q0 + t * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 32 + fq * 8;
            h8 val = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qi < Tq && c < D) val = *reinterpret_cast<const h8*>(qb + (long)qi * ldq + c);
            qf[t][ks] = val;
        }
    }

    f4 o[DT][QT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int t = 0; t < QT; ++t) o[i][t] = f4{0.f, 0.f, 0.f, 0.f};
    float mrun[QT], lrun[QT];     // running max in the scaled log2 domain; partial row sums (!ONES)
#pragma unroll
    for (int t = 0; t < QT; ++t) { mrun[t] = PRESC ? 0.f : -INFINITY; lrun[t] = 0.f; }
    // PRESC (the model pre-multiplies the query projection by log2(e)/sqrt(d)): the scores come out
    // of the QK^T MFMA in the log2 domain already, and the running reference `mrun` is subtracted by
    // starting that MFMA's accumulators at -mrun instead of 0 -- the per-score FMA of the general
    // path disappears (15 % of this VALU-issue-bound loop at d = 40).
    f4 negm[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) negm[t] = f4{0.f, 0.f, 0.f, 0.f};

    // ---- K / V tiles go global -> LDS by DMA (buffer_load ... lds, 16 B per lane, 1 KiB per wave
    //      instruction), no register staging: slot p = 64 * instr + lane of a tile is (row p / SPR, chunk
    //      p % SPR) of the padded LDS row, so a lane's source is fixed up to the tile's row offset.
    //      Pad chunks are left alone (zeros / the ones column written once above); rows past Tk read
    //      out of range and land as zeros. ----
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int SPRK = KSTR / 8, SPRV = VSTR / 8;           // 16-byte slots per padded row = DMA instrs per tile
    // The tile's SPRK + SPRV pieces are dealt round-robin over the four waves as ONE list (K pieces first):
    // with separate K and V lists wave 0 issued 4 of d = 40's 11 pieces and waves 2, 3 two each, and a
    // piece's issue (60-185 cycles, MI355X_MICROARCH.md) sits on the wave's critical path.
    constexpr int NP = SPRK + SPRV, NPW = (NP + NWV - 1) / NWV;
    const int wavu = __builtin_amdgcn_readfirstlane(wave);
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kb), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vb), 0, 0x7fffffff, 0x00020000);
    int p_row[NPW], p_off[NPW];                               // off < 0: pad chunk, lane sits out
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int q = wavu + NWV * j;                         // wave-uniform piece id
        const bool isk = q < SPRK;
        const int spr = isk ? SPRK : SPRV;
        const int p = (isk ? q : q - SPRK) * 64 + lane;
        const int r = p / spr, c = p - r * spr;
        p_row[j] = r;
        p_off[j] = c < CH ? (int)((r * (isk ? ldk : ldv) + c * 8) * 2) : -1;
    }
    auto issue_tile = [&](int bufi, int kt0) {
        half_t* dK = sK + bufi * TILE_HALVES;
        half_t* dV = sV + bufi * TILE_HALVES;
        const unsigned kbase = (unsigned)((long)kt0 * ldk * 2), vbase = (unsigned)((long)kt0 * ldv * 2);
        const int lim = Tk - kt0;                             // rows of this tile that exist
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int q = wavu + NWV * j;
            if (q < NP && p_off[j] >= 0) {
                const bool isk = q < SPRK;
                const unsigned voff = p_row[j] < lim ? (isk ? kbase : vbase) + (unsigned)p_off[j] : kOOB;
                half_t* dst = isk ? dK + q * 512 : dV + (q - SPRK) * 512;
                if (isk)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
            }
        }
    };
    if (PREFETCH) {
        __syncthreads();                       // the padding of both buffers is in place before any tile lands
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    int cur = 0;
    for (int kt0 = 0; kt0 < Tk; kt0 += KT) {
        if (PREFETCH) {
            // tile t+1 streams into the other buffer (last read one iteration ago, fenced by the barrier
            // that closed that iteration) while this one is consumed
            if (kt0 + KT < Tk) issue_tile(cur ^ 1, kt0 + KT);
        } else {
            __syncthreads();                   // previous tile fully consumed (and padding written)
            for (int idx = tid; idx < NCH; idx += NTH) {
                const int r = idx / CH, c = (idx - r * CH) * 8;
                h8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kt0 + r < Tk) {
                    kv = *reinterpret_cast<const h8*>(kb + (long)(kt0 + r) * ldk + c);
                    vv = *reinterpret_cast<const h8*>(vb + (long)(kt0 + r) * ldv + c);
                }
                *reinterpret_cast<h8*>(sK + r * KSTR + c) = kv;
                *reinterpret_cast<h8*>(sV + r * VSTR + c) = vv;
            }
            __syncthreads();
        }
        const half_t* cK = sK + cur * TILE_HALVES;
        const half_t* cV = sV + cur * TILE_HALVES;

        // ---- S^T = K Q^T : 4 key subtiles x QT query subtiles ----
        f4 s[NSUB][QT];
#pragma unroll
        for (int ksub = 0; ksub < NSUB; ++ksub)
#pragma unroll
            for (int t = 0; t < QT; ++t) s[ksub][t] = PRESC ? negm[t] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int ksub = 0; ksub < NSUB; ++ksub) {
                const h8 kf = *reinterpret_cast<const h8*>(cK + (ksub * 16 + fr) * KSTR + ks * 32 + fq * 8);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    s[ksub][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[t][ks], s[ksub][t], 0, 0, 0);
            }
        }
        if (kt0 + KT > Tk) {                   // ragged last tile: keys >= Tk never win and weigh 0
#pragma unroll
            for (int ksub = 0; ksub < NSUB; ++ksub)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (kt0 + ksub * 16 + fq * 4 + j >= Tk) {
#pragma unroll
                        for (int t = 0; t < QT; ++t) s[ksub][t][j] = -INFINITY;
                    }
        }

        if (causal && kt0 + KT - 1 > q0) {     // wave-uniform: some key of the tile may be ahead of a query
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int qi = q0 + t * 16 + fr;
#pragma unroll
                for (int ksub = 0; ksub < NSUB; ++ksub)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kt0 + ksub * 16 + fq * 4 + j > qi) s[ksub][t][j] = -INFINITY;
            }
        }

        // ---- online softmax: per query = per lane column ----
        unsigned pf[QT][NKK][4];               // P^T as packed fp16 pairs: [k-step of 32 keys][4 dwords]
        if constexpr (PRESC) {
            // s already holds (score - mrun) in log2 units.  The reference only has to keep exp2 inside
            // fp16: it moves (wave-uniformly) when some score exceeds it by 2^8, and on the first
            // tile, where it is set to the tile's own row maximum.
            float mx[QT];                      // this lane's 16 keys only: enough to decide, wave-wide,
            bool grow = kt0 == 0;              // whether anybody needs the reference moved
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float m = -INFINITY;
#pragma unroll
                for (int ksub = 0; ksub < NSUB; ++ksub)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m = fmaxf(m, s[ksub][t][j]);
                mx[t] = m;
                grow |= m > kRescaleThreshold;
            }
            if (__any(grow)) {
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 16));     // the query's row maximum over the tile
                    mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 32));
                    const float delta = kt0 == 0 ? mx[t] : fmaxf(mx[t], 0.f);
                    const float alpha = kt0 == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);   // o, l are 0 on tile 0
                    mrun[t] += delta;
                    negm[t] = f4{-mrun[t], -mrun[t], -mrun[t], -mrun[t]};
#pragma unroll
                    for (int i = 0; i < DT; ++i) o[i][t] *= alpha;
                    lrun[t] *= alpha;
#pragma unroll
                    for (int ksub = 0; ksub < NSUB; ++ksub) s[ksub][t] -= delta;
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float psum = 0.f;
#pragma unroll
                for (int ksub = 0; ksub < NSUB; ++ksub) {
                    float p[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p[j] = __builtin_amdgcn_exp2f(s[ksub][t][j]);
                        if (!ONES) psum += p[j];
                    }
                    pf[t][ksub >> 1][(ksub & 1) * 2] = pack_rtz(p[0], p[1]);
                    pf[t][ksub >> 1][(ksub & 1) * 2 + 1] = pack_rtz(p[2], p[3]);
                }
                if (!ONES) lrun[t] += psum;
            }
        } else {
            bool grow = false;
            float mx[QT];
    #pragma unroll
            for (int t = 0; t < QT; ++t) {
                float m = -INFINITY;
    #pragma unroll
                for (int ksub = 0; ksub < NSUB; ++ksub)
    #pragma unroll
                    for (int j = 0; j < 4; ++j) m = fmaxf(m, s[ksub][t][j]);
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                mx[t] = m * scale_log2e;
                grow |= mx[t] > mrun[t] + kRescaleThreshold;
            }
            if (__any(grow)) {                     // wave-uniform: rescale everything kept at the old max
    #pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const float mnew = fmaxf(mrun[t], mx[t]);
                    const float alpha = __builtin_amdgcn_exp2f(mrun[t] - mnew);
                    mrun[t] = mnew;
                    lrun[t] *= alpha;
    #pragma unroll
                    for (int i = 0; i < DT; ++i) o[i][t] *= alpha;
                }
            }
    #pragma unroll
            for (int t = 0; t < QT; ++t) {
                const float nm = -mrun[t];
                float psum = 0.f;
    #pragma unroll
                for (int ksub = 0; ksub < NSUB; ++ksub) {
                    float p[4];
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ksub][t][j], scale_log2e, nm));
                        if (!ONES) psum += p[j];
                    }
                    pf[t][ksub >> 1][(ksub & 1) * 2] = pack_rtz(p[0], p[1]);
                    pf[t][ksub >> 1][(ksub & 1) * 2 + 1] = pack_rtz(p[2], p[3]);
                }
                if (!ONES) lrun[t] += psum;
            }
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int i = 0; i < DT; ++i) {
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const half_t* a0 = cV + (kk * 32 + fq * 4 + (fr >> 2)) * VSTR + i * 16 + (fr & 3) * 4;
                const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4v*)(a0));
                const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4v*)(a0 + 16 * VSTR));
                union { struct { s4v a, b; } p; h8 v; } u;
                u.p.a = lo; u.p.b = hi;
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    union { unsigned w[4]; h8 v; } pb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pb.w[e] = pf[t][kk][e];
                    o[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u.v, pb.v, o[i][t], 0, 0, 0);
                }
            }
        }
        if (PREFETCH) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of tile t+1 has landed
            __syncthreads();     // the one barrier per tile: publishes tile t+1, retires buffer `cur`
            cur ^= 1;
        }
    }

    // ---- normalise and store: lane holds 4 consecutive d of one query ----
    half_t* ob = out + (long)b * Tq * ldo + h * D;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l;
        if (ONES) {
            // row D of O^T = sum_k P: held by lane group (D % 16) / 4, register D % 4
            constexpr int LT = D / 16, LQ = (D % 16) / 4, LJ = D % 4;
            l = __shfl(o[LT][t][LJ], LQ * 16 + fr);
        } else {
            l = lrun[t];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = 1.0f / l;
        const int qi = q0 + t * 16 + fr;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const int dd = i * 16 + fq * 4;
            if (qi < Tq && dd < D) {
                const f4 val = o[i][t] * inv;
                h4 w = {(half_t)val[0], (half_t)val[1], (half_t)val[2], (half_t)val[3]};
                *reinterpret_cast<h4*>(ob + (long)qi * ldo + dd) = w;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int D, int QT, int KT, bool PRESC, int NWV = 4>
int launch_attn(const half_t* q, const half_t* k, const half_t* v, half_t* out, int B, int Tq, int Tk,
                int heads, long ldq, long ldk, long ldv, long ldo, int causal, bool q_has_scale, hipStream_t s) {
    constexpr int DK = (D + 31) / 32 * 32;
    constexpr int DT = (D + 15) / 16 + ((D % 16 == 0 && D <= 64) ? 1 : 0);
    constexpr size_t lds = (size_t)KT * (k_row_halves(D) * 2 + odd32_bytes(DT * 16 * 2)) * (D <= 160 ? 2 : 1);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<D, QT, KT, PRESC, NWV>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // q_has_scale with the general kernel: the scores only need the running-max subtraction
    const float scale_log2e = q_has_scale ? 1.0f : 1.4426950408889634f / sqrtf((float)D);
    const int qblocks = cdiv(Tq, 16 * QT * NWV);
    hipLaunchKernelGGL((attn_kernel<D, QT, KT, PRESC, NWV>), dim3(qblocks * B * heads), dim3(64 * NWV), lds, s, q, k, v,
                       out, Tq, Tk, heads, ldq, ldk, ldv, ldo, scale_log2e, causal, qblocks);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

bool attention_supported(int d) {
    return d == 32 || d == 40 || d == 64 || d == 80 || d == 128 || d == 160 || d == 512;
}

int launch_attention(const half_t* q, const half_t* k, const half_t* v, half_t* out, int B, int Tq,
                     int Tk, int heads, int d, long ldq, long ldk, long ldv, long ldo, hipStream_t s, int causal,
                     int prescaled) {
    if ((ldq | ldk | ldv | ldo) % 8 != 0) { set_error("attention: row strides must be multiples of 8"); return 1; }
    if (Tk <= 0 || Tq <= 0) return 0;
    // FAST: whether the accumulator-start form pays for this head dim (measured with the DMA staging:
    // -11 % at d = 40, where the loop is VALU-issue-bound; a wash or slightly worse from d = 64 up);
    // otherwise pre-scaled queries run the general kernel with a unit scale
#define SD_ATTN_CASE(DD, QQ, KK, FAST) \
    case DD: return (prescaled && FAST) \
        ? launch_attn<DD, QQ, KK, FAST>(q, k, v, out, B, Tq, Tk, heads, ldq, ldk, ldv, ldo, causal, true, s) \
        : launch_attn<DD, QQ, KK, false>(q, k, v, out, B, Tq, Tk, heads, ldq, ldk, ldv, ldo, causal, prescaled != 0, s)
    // Few, long query blocks leave CUs idle on the small maps (16 x 16 latents: 256 queries x 64 (batch, head)
    // pairs = 128 blocks of 128 queries for 256 CUs): halve the block there.  (d = 40 was also tried at 48 / 32 /
    // 16 queries per wave for more waves per SIMD: 299 -> 318 / 318 / 403 us on the 4096-token case: it is not
    // latency-bound.)
    // eight waves per block where there are enough query blocks to fill the chip with them (SD_ATTN_NWV=4: A/B switch)
    static const int nwv = getenv("SD_ATTN_NWV") ? atoi(getenv("SD_ATTN_NWV")) : 8;
    if (d == 40 && prescaled && !causal && nwv == 8 && (long)cdiv(Tq, 512) * B * heads >= 512)
        return launch_attn<40, 4, 64, true, 8>(q, k, v, out, B, Tq, Tk, heads, ldq, ldk, ldv, ldo, causal, true, s);
    // d = 80 (the 32 x 32 level): 256 queries per block where that still gives one block per CU -- 42.5 -> 36.9 us on the
    // 1024-token self-attention; d = 64 (SDXL) measured slower with eight waves (259.6 -> 269.8, 42.5 -> 48.3 us) and stays at four
    static const int nwv80 = getenv("SD_ATTN_NWV80") ? atoi(getenv("SD_ATTN_NWV80")) : 8;
    if (nwv80 == 8 && !causal && d == 80 && (long)cdiv(Tq, 256) * B * heads >= 256)
        return launch_attn<80, 2, 64, false, 8>(q, k, v, out, B, Tq, Tk, heads, ldq, ldk, ldv, ldo, causal, prescaled != 0, s);
    if (d == 160 && (long)cdiv(Tq, 128) * B * heads < 256)
        return launch_attn<160, 1, 64, false>(q, k, v, out, B, Tq, Tk, heads, ldq, ldk, ldv, ldo, causal, prescaled != 0, s);
    switch (d) {
        SD_ATTN_CASE(32, 2, 64, true);
        SD_ATTN_CASE(40, 4, 64, true);
        SD_ATTN_CASE(64, 2, 64, false);
        SD_ATTN_CASE(80, 2, 64, false);
        SD_ATTN_CASE(128, 2, 64, false);
        SD_ATTN_CASE(160, 2, 64, false);
        SD_ATTN_CASE(512, 1, 64, false);
        default:
            set_error("attention: unsupported head dim " + std::to_string(d));
            return 4;
    }
#undef SD_ATTN_CASE
}

}  // namespace sd
