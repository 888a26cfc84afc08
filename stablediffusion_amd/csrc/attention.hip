// Flash-style attention for gfx950 (CDNA4, wave64): out = softmax(q k^T / sqrt(d)) v, fp16 in/out,
// fp32 scores / running max / running sum / output accumulators; the T x T score matrix is never
// materialised.  Replaces F.scaled_dot_product_attention under diffusers' AttnProcessor2_0 for the
// UNet self/cross attention and the VAE mid-block attention (call sites
// /root/reference/pipelines/sd_unified_pipeline.py:475-482, :523).
//
// MI355X-specific structure:
//   * "swapped" products so every softmax quantity of a query lives in one lane:
//       S^T = K Q^T   (MFMA A = K rows from LDS, B = Q fragments held in registers)
//       O^T = V^T P^T (MFMA A = V^T fetched with ds_read_b64_tr_b16 from a row-major V tile,
//                      B = P^T taken straight from the S^T accumulators -- no LDS round trip)
//     A 16x16 accumulator tile has the query on lane&15 and 4 keys per lane-group, which is
//     exactly the k-slot layout of the B operand once the V^T fragment is fetched in the same
//     permuted key order.
//   * K / V tiles of 64 keys in LDS with row strides that are odd multiples of 32 bytes:
//     conflict-free for both ds_read_b128 fragment reads and the transposed reads.
//   * head dims 40 / 80 / 160 (SD1.5), 64 (SDXL), 512 (VAE), 32 (test configs); the QK^T
//     contraction is zero-padded to a multiple of 32, the PV row tiles to a multiple of 16.
#include "kernels.h"

namespace sd {
namespace {

constexpr int KT = 64;  // keys per tile

constexpr int odd32_bytes(int bytes) { return ((((bytes + 31) / 32) | 1)) * 32; }

template <int D, int QT>
__global__ __launch_bounds__(256) void attn_kernel(const half_t* __restrict__ q,
                                                   const half_t* __restrict__ k,
                                                   const half_t* __restrict__ v,
                                                   half_t* __restrict__ out, int Tq, int Tk,
                                                   int heads, long ldq, long ldk, long ldv,
                                                   long ldo, float scale_log2e) {
    constexpr int DK = (D + 31) / 32 * 32;
    constexpr int KS = DK / 32;
    constexpr int DT = (D + 15) / 16;
    constexpr int KSTR = odd32_bytes(DK * 2) / 2;        // halves
    constexpr int VSTR = odd32_bytes(DT * 16 * 2) / 2;   // halves
    constexpr int KCH = KSTR / 8, VCH = VSTR / 8;        // 16-byte chunks per LDS row
    constexpr int QB = 64 * QT;                          // queries per block

    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sK = reinterpret_cast<half_t*>(smem);
    half_t* sV = sK + KT * KSTR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / heads, h = bh - b * heads;
    const int q0 = blockIdx.x * QB + wave * (16 * QT);

    const half_t* qb = q + (long)b * Tq * ldq + h * D;
    const half_t* kb = k + (long)b * Tk * ldk + h * D;
    const half_t* vb = v + (long)b * Tk * ldv + h * D;

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
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { mrun[t] = -INFINITY; lrun[t] = 0.f; }

    for (int kt0 = 0; kt0 < Tk; kt0 += KT) {
        __syncthreads();
        // ---- stage K and V tiles (zero-filled padding) ----
        for (int idx = tid; idx < KT * KCH; idx += 256) {
            const int r = idx / KCH, c = (idx - r * KCH) * 8;
            h8 val = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kt0 + r < Tk && c < D) val = *reinterpret_cast<const h8*>(kb + (long)(kt0 + r) * ldk + c);
            *reinterpret_cast<h8*>(sK + r * KSTR + c) = val;
        }
        for (int idx = tid; idx < KT * VCH; idx += 256) {
            const int r = idx / VCH, c = (idx - r * VCH) * 8;
            h8 val = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kt0 + r < Tk && c < D) val = *reinterpret_cast<const h8*>(vb + (long)(kt0 + r) * ldv + c);
            *reinterpret_cast<h8*>(sV + r * VSTR + c) = val;
        }
        __syncthreads();

        // ---- S^T = K Q^T : 4 key subtiles x QT query subtiles ----
        f4 s[4][QT];
#pragma unroll
        for (int ksub = 0; ksub < 4; ++ksub)
#pragma unroll
            for (int t = 0; t < QT; ++t) s[ksub][t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int ksub = 0; ksub < 4; ++ksub) {
                const h8 kf = *reinterpret_cast<const h8*>(sK + (ksub * 16 + fr) * KSTR + ks * 32 + fq * 8);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    s[ksub][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[t][ks], s[ksub][t], 0, 0, 0);
            }
        }

        // ---- online softmax (per query = per lane column), scores scaled into log2 domain ----
        const bool tail = kt0 + KT > Tk;
        h8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -INFINITY;
#pragma unroll
            for (int ksub = 0; ksub < 4; ++ksub)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float val = s[ksub][t][j] * scale_log2e;
                    if (tail && kt0 + ksub * 16 + fq * 4 + j >= Tk) val = -INFINITY;
                    s[ksub][t][j] = val;
                    mx = fmaxf(mx, val);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[t], mx);
            const float alpha = exp2f(mrun[t] - mnew);
            mrun[t] = mnew;
            float psum = 0.f;
#pragma unroll
            for (int ksub = 0; ksub < 4; ++ksub)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pv = exp2f(s[ksub][t][j] - mnew);
                    psum += pv;
                    pf[t][ksub >> 1][(ksub & 1) * 4 + j] = (half_t)pv;
                }
            lrun[t] = lrun[t] * alpha + psum;
#pragma unroll
            for (int i = 0; i < DT; ++i) o[i][t] *= alpha;
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int i = 0; i < DT; ++i) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half_t* a0 = sV + (kk * 32 + fq * 4 + (fr >> 2)) * VSTR + i * 16 + (fr & 3) * 4;
                const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4v*)(a0));
                const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4v*)(a0 + 16 * VSTR));
                union { struct { s4v a, b; } p; h8 v; } u;
                u.p.a = lo; u.p.b = hi;
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    o[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u.v, pf[t][kk], o[i][t], 0, 0, 0);
            }
        }
    }

    // ---- normalise and store: lane holds 4 consecutive d of one query ----
    half_t* ob = out + (long)b * Tq * ldo + h * D;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = lrun[t];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
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
}

template <int D, int QT>
int launch_attn(const half_t* q, const half_t* k, const half_t* v, half_t* out, int B, int Tq, int Tk,
                int heads, long ldq, long ldk, long ldv, long ldo, hipStream_t s) {
    constexpr int DK = (D + 31) / 32 * 32;
    constexpr int DT = (D + 15) / 16;
    constexpr size_t lds = (size_t)KT * (odd32_bytes(DK * 2) + odd32_bytes(DT * 16 * 2));
    static bool attr_set = false;
    if (!attr_set) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<D, QT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
    hipLaunchKernelGGL((attn_kernel<D, QT>), dim3(cdiv(Tq, 64 * QT), B * heads), dim3(256), lds, s, q, k, v,
                       out, Tq, Tk, heads, ldq, ldk, ldv, ldo, scale_log2e);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

bool attention_supported(int d) {
    return d == 32 || d == 40 || d == 64 || d == 80 || d == 128 || d == 160 || d == 512;
}

int launch_attention(const half_t* q, const half_t* k, const half_t* v, half_t* out, int B, int Tq,
                     int Tk, int heads, int d, long ldq, long ldk, long ldv, long ldo, hipStream_t s) {
    if ((ldq | ldk | ldv | ldo) % 8 != 0) { set_error("attention: row strides must be multiples of 8"); return 1; }
    if (Tk <= 0 || Tq <= 0) return 0;
#define SD_ATTN_CASE(DD, QQ) \
    case DD: return launch_attn<DD, QQ>(q, k, v, out, B, Tq, Tk, heads, ldq, ldk, ldv, ldo, s)
    switch (d) {
        SD_ATTN_CASE(32, 2);
        SD_ATTN_CASE(40, 2);
        SD_ATTN_CASE(64, 2);
        SD_ATTN_CASE(80, 2);
        SD_ATTN_CASE(128, 2);
        SD_ATTN_CASE(160, 2);
        SD_ATTN_CASE(512, 1);
        default:
            set_error("attention: unsupported head dim " + std::to_string(d));
            return 4;
    }
#undef SD_ATTN_CASE
}

}  // namespace sd
