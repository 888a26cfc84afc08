// Implicit-GEMM convolution / linear for gfx950 (CDNA4): NHWC fp16 activations, fp16 packed
// weights [Cout][KH][KW][Cin], fp32 accumulation on v_mfma_f32_16x16x32_f16.
//
// Replaces the aten conv2d / linear calls diffusers issues under UNet2DConditionModel.forward and
// AutoencoderKL.decode (call sites /root/reference/pipelines/sd_unified_pipeline.py:475-482, :523).
//
// Design (MI355X-first, not a cuDNN-shaped port):
//   * no im2col: each 64-wide K slab of a tile row is one (kh,kw) tap x 64 input channels, which is
//     contiguous in NHWC, so the A tile is gathered with 16-byte loads; padding, stride-2 and the
//     nearest-2x upsample are pure address arithmetic.
//   * 256 threads = 4 wave64; block tile BM x BN x 64, wave tile (BM/WM) x (BN/WN) built from
//     16x16x32 MFMAs.  Weights are the MFMA "A" operand and pixels the "B" operand, so a lane ends
//     up with 4 consecutive output channels of one pixel (8-byte packed epilogue writes).
//   * LDS tiles are [rows][64 halves] with the 16-byte chunk index XOR-ed with (row & 7): both the
//     ds_write_b128 staging writes and the ds_read_b128 fragment reads are bank-conflict free.
//   * register-staged double buffering: global loads for slab t+1 are in flight while slab t is
//     multiplied; one barrier per slab.
//   * epilogue through LDS: bias + per-sample row add (time embedding) in fp32, then 16-byte
//     coalesced stores with the residual add or the GEGLU gate fused.
//   * block ids are remapped so each XCD's L2 sees a contiguous range of tiles.
#include "kernels.h"

namespace sd {

namespace {

constexpr int BK = 64;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmParams p) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int LDC = BN + 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sA = reinterpret_cast<half_t*>(smem);  // [2][BM*BK]
    half_t* sB = sA + 2 * BM * BK;                 // [2][BN*BK]
    half_t* sC = reinterpret_cast<half_t*>(smem);  // epilogue overlay [BM][LDC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread gather coordinates (fixed rows, fixed 16B chunk) ----
    const int c8 = tid & 7;
    const int r0 = tid >> 3;
    const int OHW = p.OH * p.OW;
    const int IH = p.H << p.up, IW = p.W << p.up;
    int a_ih[A_IT], a_iw[A_IT];
    long a_pix[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if (p.KS == 1 && p.stride == 1 && p.up == 0) {
            a_ih[i] = 0; a_iw[i] = 0; a_pix[i] = mm;
        } else {
            const int n = mm / OHW;
            const int rem = mm - n * OHW;
            const int oh = rem / p.OW;
            const int ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pad;
            a_iw[i] = ow * p.stride - p.pad;
            a_pix[i] = (long)n * p.H * p.W;
        }
    }
    const half_t* wrow[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
        wrow[i] = p.w + (long)(n0 + r0 + 32 * i) * p.K + c8 * 8;

    h8 ra[A_IT], rb[B_IT];
    int kh = 0, kw = 0, ci0 = 0;   // tap / channel offset of the NEXT slab to load
    const bool pointwise = (p.KS == 1 && p.stride == 1 && p.up == 0);

    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pointwise) {
                if (a_ok[i]) v = *reinterpret_cast<const h8*>(p.x + a_pix[i] * p.ldx + k0 + c8 * 8);
            } else {
                int ih = a_ih[i] + kh, iw = a_iw[i] + kw;
                const bool ok = a_ok[i] && ih >= 0 && ih < IH && iw >= 0 && iw < IW;
                ih >>= p.up; iw >>= p.up;
                if (ok)
                    v = *reinterpret_cast<const h8*>(
                        p.x + (a_pix[i] + (long)ih * p.W + iw) * p.ldx + ci0 + c8 * 8);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = *reinterpret_cast<const h8*>(wrow[i] + k0);
        // K order = [Cin/64][KH][KW][64] (misc.hip pack_conv_kernel): taps innermost
        if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ci0 += BK; } }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<h8*>(sA + buf * BM * BK + r * BK + ((c8 ^ (r & 7)) << 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<h8*>(sB + buf * BN * BK + r * BK + ((c8 ^ (r & 7)) << 3)) = rb[i];
        }
    };

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    load_slab(0);
    store_slab(0);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_slab((kt + 1) * BK);
        const half_t* cA = sA + buf * BM * BK;
        const half_t* cB = sB + buf * BN * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 fa[TM], fb[TN];
            const int ch = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * WTM + i * 16 + fr;
                fa[i] = *reinterpret_cast<const h8*>(cA + r * BK + ((ch ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WTN + j * 16 + fr;
                fb[j] = *reinterpret_cast<const h8*>(cB + r * BK + ((ch ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc (+bias, +rowadd) -> fp16 -> LDS tile [pixel][channel] ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pr = wm * WTM + i * 16 + fr;
        const int m = m0 + pr;
        int nimg = 0;
        if (p.rowadd) nimg = (m < p.M ? m : 0) / OHW;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * WTN + j * 16 + fq * 4;
            f4 v = acc[i][j];
            if (p.bias) {
                const f4 b = *reinterpret_cast<const f4*>(p.bias + n0 + col);
                v += b;
            }
            if (p.rowadd && n0 + col < p.Cout) {
                const f4 t = *reinterpret_cast<const f4*>(p.rowadd + (long)nimg * p.rowadd_ld + n0 + col);
                v += t;
            }
            h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<h4*>(sC + pr * LDC + col) = o;
        }
    }
    __syncthreads();

    if (p.geglu) {
        constexpr int OCH = BN / 16;                // 16-byte output chunks per tile row
        const int out_n0 = n0 >> 1;
        for (int idx = tid; idx < BM * OCH; idx += 256) {
            const int r = idx / OCH, oc = (idx - r * OCH) * 8;
            const int m = m0 + r;
            const int hcol = (oc >> 6) * 128 + (oc & 63);
            if (m < p.M && n0 + hcol < p.Cout) {
                const h8 hv = *reinterpret_cast<const h8*>(sC + r * LDC + hcol);
                const h8 gv = *reinterpret_cast<const h8*>(sC + r * LDC + hcol + 64);
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)hv[e] * gelu_erf_f((float)gv[e]));
                *reinterpret_cast<h8*>(p.y + (long)m * p.ldy + out_n0 + oc) = o;
            }
        }
        return;
    }

    constexpr int CH = BN / 8;
    if ((p.Cout & 7) == 0) {
        for (int idx = tid; idx < BM * CH; idx += 256) {
            const int r = idx / CH, c = (idx - r * CH) * 8;
            const int m = m0 + r, n = n0 + c;
            if (m < p.M && n < p.Cout) {
                h8 v = *reinterpret_cast<const h8*>(sC + r * LDC + c);
                if (p.res) {
                    const h8 rv = *reinterpret_cast<const h8*>(p.res + (long)m * p.ldres + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
                }
                *reinterpret_cast<h8*>(p.y + (long)m * p.ldy + n) = v;
            }
        }
    } else {
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int r = idx / BN, c = idx - r * BN;
            const int m = m0 + r, n = n0 + c;
            if (m < p.M && n < p.Cout) {
                float v = (float)sC[r * LDC + c];
                if (p.res) v += (float)p.res[(long)m * p.ldres + n];
                p.y[(long)m * p.ldy + n] = (half_t)v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const IGemmParams& p, hipStream_t s) {
    constexpr size_t stage = (size_t)2 * (BM + BN) * BK * sizeof(half_t);
    constexpr size_t epi = (size_t)BM * (BN + 8) * sizeof(half_t);
    constexpr size_t lds = stage > epi ? stage : epi;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WM, WN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int tiles = cdiv(p.M, BM) * cdiv(p.Cout, BN);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN>), dim3(tiles), dim3(256), lds, s, p);
    SD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

// 0: 128x128, 1: 128x64, 2: 64x64
static int pick_tile(const IGemmParams& p) {
    // 128x128 by default; narrower tiles when that removes column padding waste or when the
    // 128x128 grid would leave most of the 256 CUs idle.
    const long t128 = (long)cdiv(p.M, 128) * cdiv(p.Cout, 128);
    const bool waste128 = (p.Cout % 128) != 0 && (p.Cout % 128) <= 64;
    if (!p.geglu && (waste128 || t128 < 192)) {
        const long t64 = (long)cdiv(p.M, 64) * cdiv(p.Cout, 64);
        if (t128 < 96 && t64 >= t128 * 2) return 2;
        return 1;
    }
    return 0;
}

const char* igemm_variant(const IGemmParams& p) {
    static const char* names[3] = {"igemm_kernel<128,128>", "igemm_kernel<128,64>", "igemm_kernel<64,64>"};
    return names[pick_tile(p)];
}

int launch_igemm(const IGemmParams& p, hipStream_t s) {
    if (p.K % BK != 0 || p.Cin % BK != 0) {
        set_error("igemm: K and Cin must be multiples of 64 (pad small-channel inputs via im2col)");
        return 1;
    }
    if (p.geglu && (p.Cout % 128 != 0)) {
        set_error("igemm: GEGLU needs Cout % 128 == 0");
        return 1;
    }
    if (p.M <= 0 || p.Cout <= 0) return 0;
    switch (pick_tile(p)) {
        case 2: return launch_cfg<64, 64, 2, 2>(p, s);
        case 1: return launch_cfg<128, 64, 2, 2>(p, s);
        default: return launch_cfg<128, 128, 2, 2>(p, s);
    }
}

}  // namespace sd
