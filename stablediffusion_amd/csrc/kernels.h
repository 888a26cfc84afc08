// Host-side launchers of the HIP kernels (definitions in *.hip).  All tensors are device memory,
// activations are NHWC fp16 "views": a base pointer plus a row stride (`ld`, in elements) so a
// tensor can live inside a wider buffer (zero-copy skip concatenation).
#pragma once
#include "common.h"

namespace sd {

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution / linear (igemm.hip)
//   y[m, co] = bias[co] + rowadd[n(m), co] + res[m, co] + sum_k A[m, k] * Wp[co, k]
//   m = (n, oh, ow);  k = (kh, kw, ci);  A gathered on the fly from x (zero padding, stride,
//   optional nearest-2x upsample of the input).  Wp is the packed weight [rows >= Cout][K].
// ---------------------------------------------------------------------------------------------
struct IGemmParams {
    const half_t* x; long ldx;
    const half_t* w;
    const float* bias;
    const float* rowadd; int rowadd_ld;
    const half_t* res; long ldres;
    half_t* y; long ldy;
    int N, H, W, Cin;       // input geometry (before the optional 2x upsample)
    int OH, OW, Cout;       // output geometry; Cout = GEMM columns (2x the stored width for GEGLU)
    int KS, stride, pad, up;
    int M, K;
    int geglu;              // 1: y[m, j] = hidden_j * gelu(gate_j), weights interleaved per 64
    // Block -> tile order.  Each XCD (own 4 MB L2) works through a contiguous range of block ids; 0: the
    // N tiles of one M tile are neighbours (the activation rows are fetched into one L2, every L2 pulls
    // all the weights), 1: the M tiles of one N tile are neighbours (each weight panel goes to one L2,
    // the activations to all).  Set by the launcher: 1 when the weights outweigh the activations.
    int mfast = 0;
    int act = 0;            // activation after bias, before residual (LDS-DMA kernels, no split-K):
                            // 0 none, 1 quick_gelu x*sigmoid(1.702x) (CLIP-L MLP), 2 exact-erf gelu (OpenCLIP MLP)
    // ---- LayerNorm folded across the GEMM (LDS-DMA kernels, pointwise, no split-K) ----
    // Producer side: the epilogue also writes, per output row and per column tile of this launch, the sum
    // and the sum of squares of the fp16 values it stores: rowstat_out[(m * rowstat_parts + tile_n) * 2 + {0,1}].
    float* rowstat_out = nullptr;
    int rowstat_parts = 0;          // = column tiles of the launch (filled in by the launcher)
    // Consumer side: y = LN(x) W^T + b with LN's affine folded into the weights at pack time
    // (W' = W diag(gamma), b' = b + W beta, wsum_n = sum_k W'_nk) becomes
    //     y[m, n] = rstd_m * (acc[m, n] - mean_m * wsum[n]) + b'[n],   acc = x W'^T on the raw x
    // with mean / rstd of row m from the producer's partial sums (ln_parts of them, over ln_C channels).
    const float* ln_stat = nullptr;
    int ln_parts = 0, ln_C = 0;
    float ln_eps = 0.f;
    const float* ln_wsum = nullptr;
    // ---- GroupNorm statistics of the output, for the GroupNorm that follows (LDS-DMA kernels, no split-K,
    //      every tile inside one image, group boundaries on tile boundaries: igemm2_emits_gnstats) ----
    // gnstat_out[((img * tiles_per_image + tile) * gn_groups + g) * 2 + {0,1}] = (mean, M2) of the tile's rows
    // x the group's channels (the GnStats layout of launch_groupnorm)
    float* gnstat_out = nullptr;
    int gn_groups = 0;
    // ---- GroupNorm (+ SiLU) of the INPUT, applied inside the convolution (conv3x3_halo_kernel<BN, true>: the halo
    //      tile of every 64-channel slab is normalised in LDS after its DMA has landed; igemm2_gn_fusable) ----
    // gni_part: (mean, M2) summaries of x in the GnStats layout, gni_S of them per image over gni_rows pixels each;
    // gni_gb: the norm's affine packed per 64 channels [Cin / 64][64 gamma | 64 beta] (NormW::gb).
    const float* gni_part = nullptr;
    int gni_S = 0;
    long gni_rows = 0;
    int gni_groups = 0;
    float gni_eps = 0.f;
    int gni_silu = 0;
    const float* gni_gb = nullptr;
    int dbg_unchecked = 0;          // tests: wsgemm's residual descriptor without its range check
    // ---- y = acc_scale * (x W^T) + bias_scale * bias (+ rowadd + res): the VAE encoder's range-scaled form (LDS-DMA
    //      kernels and their split-K reductions; igemm2_scales_ok).  Scales are powers of two.  See VAE::run_encode. ----
    float acc_scale = 1.f, bias_scale = 1.f;
};
// Whether launch_igemm2 can apply a GroupNorm of `groups` groups to this problem's input (p.gni_* set by the caller).
bool igemm2_gn_fusable(const IGemmParams& p, int groups);
// Whether launch_igemm2 honours acc_scale / bias_scale for this problem (not the GEGLU / weight-stationary forms)
bool igemm2_scales_ok(const IGemmParams& p);
// Whether launch_igemm2 will honour p.gnstat_out for `groups` groups; *rows = pixels per tile (GnStats::rows).
bool igemm2_emits_gnstats(const IGemmParams& p, int groups, int* rows);
// Whether launch_igemm2 will honour p.rowstat_out for this problem (LDS-DMA kernel, no split-K); when not,
// the caller runs launch_row_stats on the output instead.  Fills *parts with the column-tile count.
bool igemm2_emits_rowstats(const IGemmParams& p, int* parts);
int launch_igemm(const IGemmParams& p, hipStream_t s);
const char* igemm_variant(const IGemmParams& p);   // name of the tile variant launch_igemm picks
// LDS-DMA pipeline variants (igemm2.hip); falls back to launch_igemm when !igemm2_supported().
bool igemm2_supported(const IGemmParams& p);
void igemm2_pick(const IGemmParams& p, int* variant, int* splits);
const char* igemm2_name(int variant);
long igemm2_partial_floats(const IGemmParams& p);      // fp32 workspace needed for split-K (0 if none)
int launch_igemm2(const IGemmParams& p, float* partial, hipStream_t s);
void igemm2_force(int variant, int splits);            // tuner / tests: -1 restores the heuristic
// Pointwise 128 x 80 tile with the activation operand fetched straight into registers (igemm3.hip): launch_igemm2 takes
// it instead of the 128 x 80 LDS-DMA variants when the problem allows (no split-K, GEGLU, row add, GroupNorm summaries).
bool igemm3_supported(const IGemmParams& p);
int launch_igemm3(const IGemmParams& p, int mfast, hipStream_t s);
// Rows the packed weight matrix must be padded to (zero rows), so tile loads need no masks.
constexpr int kWeightRowPad = 256;
// Weight-stationary persistent GEMM for the K = 320 pointwise problems of the 64x64 level (wsgemm.hip):
// igemm2 variants 13 (plain, 128x160) and 14 (GEGLU, 128x128); launch_igemm2 routes to it.
bool wsgemm_supported(const IGemmParams& p);
int wsgemm_rowstat_parts(const IGemmParams& p);         // column partials per row it writes to rowstat_out
int launch_wsgemm(const IGemmParams& p, hipStream_t s);
// Persistent GEGLU projection for the K >= 128 pointwise problems with >= 512 tiles of 256 x 128 (pgemm.hip): one block
// per CU owns an M tile and a run of N tiles, its LDS-DMA ring runs across the tile boundaries; launch_igemm2 routes to it.
bool pgemm_geglu_supported(const IGemmParams& p);
int launch_pgemm_geglu(const IGemmParams& p, hipStream_t s);

// Fused GEGLU feed-forward (ffn.hip): out = x + GEGLU(LN(x) W1^T + b1) W2^T + b2 with the 4C-wide hidden tensor kept in
// LDS / registers (C = 320: the 64 x 64 level of the SD1.5 UNet).  W1 / b1 / wsum1 are the LayerNorm-folded, GEGLU-packed
// projection (WeightStore::pack_geglu + fold_ln), W2 / b2 the packed output linear; ln_stat the row statistics of x.
struct FfnParams {
    const half_t* x; long ldx;          // [M, C]: input of the block and its residual
    half_t* y; long ldy;                // [M, C]
    const half_t* w1; const float* b1; const float* wsum1; int w1_rows;     // [w1_rows >= 8C][C]
    const half_t* w2; const float* b2; int w2_rows;                          // [w2_rows >= 384][4C]
    const float* ln_stat; int ln_parts; float ln_eps;
    int M, C, hidden;
};
bool ffn_fused_supported(const FfnParams& p);
int launch_ffn_fused(const FfnParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Normalisation (norm.hip)
// ---------------------------------------------------------------------------------------------
// GroupNorm over NHWC: x [N, HW, C] (ld), groups G.  `scratch` must hold gn_scratch_floats().
long gn_scratch_floats(int N, long HW, int C, int G);
// Statistics some earlier kernel already computed for x: per (sample, slab of `rows` consecutive pixels,
// group) the pair (mean, M2 = sum (x - mean)^2), part[((n * S + slab) * G + g) * 2 + {0,1}].  Written by
// a convolution's epilogue (IGemmParams::gnstat_out); launch_groupnorm then skips its own pass over x.
struct GnStats {
    const float* part = nullptr;
    int S = 0;
    long rows = 0;
};
// false for the small maps the single-kernel GroupNorm handles (it reads x once anyway)
bool gn_wants_stats(long HW, int C, int G);
int launch_groupnorm(const half_t* x, long ldx, const float* gamma, const float* beta,
                     half_t* y, long ldy, int N, long HW, int C, int G, float eps, int silu,
                     float* scratch, hipStream_t s, const GnStats* pre = nullptr);
// Statistics pass alone: (mean, M2) summaries of x into `scratch` (gn_scratch_floats), described by *st.
int launch_gn_stats(const half_t* x, long ldx, int N, long HW, int C, int G, float* scratch, GnStats* st, hipStream_t s);
// Merges the st.S summaries per image into one (into `out`, N * G * 2 floats) and rewrites *st to describe that.
int launch_gn_finalize(GnStats* st, float* out, int N, long HW, int C, int G, hipStream_t s);
// GroupNorm over a channel concatenation [A | B] from its two producers' summaries (norm.hip: gn_cat_finalize_kernel):
// sa / sb describe summaries over Ga / Gb sub-groups of the Ca / Cb channels; out (N * G * 2 floats) receives one
// (mean, M2) per image and group of the concatenation = GnStats{out, 1, HW}.  gn_cat_unit: the sub-group width A's
// producer has to use; B's own width must divide both the channels per group and Ca.
int gn_cat_unit(int Ca, int Cb, int G);
int launch_gn_cat_finalize(const GnStats& sa, int Ga, int Ca, const GnStats& sb, int Gb, int Cb, float* out, int N, long HW, int G,
                           hipStream_t s);
int launch_layernorm(const half_t* x, long ldx, const float* gamma, const float* beta,
                     half_t* y, long ldy, long rows, int C, float eps, hipStream_t s);
// stat[m * 2 + {0,1}] = sum, sum of squares of row m (the one-part form of IGemmParams::rowstat_out)
int launch_row_stats(const half_t* x, long ldx, float* stat, long rows, int C, hipStream_t s);
// Pack-time LayerNorm fold of a [rows][K] fp16 weight matrix, in place (rows < rows_scaled are also
// multiplied by row_scale: the attention query pre-scale):  w[n][k] <- fp16(w[n][k] * gamma[k] * s_n),
// wsum[n] = sum_k of the ROUNDED new row, bias[n] <- s_n * (bias_in[n] + sum_k w_old[n][k] * beta[k]).
// bias_in may be null (no bias); bias / wsum are fp32 arrays of `rows` entries.
int launch_ln_fold(half_t* w, long K, int rows, const float* gamma, const float* beta, const float* bias_in,
                   float* bias, float* wsum, int rows_scaled, float row_scale, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Attention (attention.hip): out = softmax(q k^T / sqrt(d)) v per (batch, head)
// ---------------------------------------------------------------------------------------------
int launch_attention(const half_t* q, const half_t* k, const half_t* v, half_t* out,
                     int B, int Tq, int Tk, int heads, int d,
                     long ldq, long ldk, long ldv, long ldo, hipStream_t s, int causal = 0, int prescaled = 0);
// prescaled = 1: q already carries log2(e)/sqrt(d) (the UNet folds it into its query projections at
// pack time, launch_scale_f16), which lets the kernel drop its per-score scale-and-subtract FMA.
bool attention_supported(int d);

// ---------------------------------------------------------------------------------------------
// Small / elementwise kernels (misc.hip)
// ---------------------------------------------------------------------------------------------
// out[b, :] = [cos(t_b f_i) | sin(t_b f_i)] (flip) or [sin | cos]; f_i = exp(-ln(1e4) i/(half-shift))
int launch_timestep_sinusoid(const float* t, int t_stride, float* out, int count, int dim, int flip, float shift,
                             long out_ld, hipStream_t s);
// y[b, n] = bias[n] + sum_k act(x[b, k]) * W[n, k]  for small b (time-embedding MLPs); fp32 in/out.
int launch_small_linear(const float* x, long ldx, const half_t* w, const float* bias, float* y,
                        long ldy, int B, int K, int Nout, int silu_in, int silu_out, hipStream_t s);
int launch_add_f32(float* y, const float* x, long n, int silu, hipStream_t s);   // y = act(y + x), x may be null
int launch_f16_to_f32(const half_t* x, float* y, long n, hipStream_t s);
int launch_f32_to_f16(const float* x, half_t* y, long n, hipStream_t s);
// NCHW fp16 [N,C,H,W] -> im2col rows [N*H*W, Kpad] for a 3x3 pad-1 conv with tiny C (k=(kh,kw,c)).
int launch_im2col_nchw3x3(const half_t* x, half_t* col, int N, int C, int H, int W, int Kpad,
                          hipStream_t s);
// NHWC [M, C] (ld) -> NCHW [N, C, HW]
int launch_nhwc_to_nchw(const half_t* x, long ldx, half_t* y, int N, long HW, int C, hipStream_t s);
// NCHW [N,C,HW] -> NHWC [M, C] (ld)
int launch_nchw_to_nhwc(const half_t* x, half_t* y, long ldy, int N, long HW, int C, hipStream_t s);
// Pointwise CxC conv on NCHW with tiny C (VAE post_quant_conv / quant_conv).
int launch_pointwise_nchw(const half_t* x, const half_t* w, const float* bias, half_t* y, int N,
                          int Cin, int Cout, long HW, hipStream_t s);
// 3x3 / stride 1 / pad 1 conv to Cout <= 4 channels (conv_out): x NHWC f16 (row stride ldx), packed weights
// [>= Cout rows][K = 9 * Cin] (rows beyond Cout are the pack's zero padding), fp32 bias; output NCHW f16.
// Worth it from about 128 K output pixels (512-pixel blocks: fewer leave most CUs idle; measured on C2: the
// 32 K-pixel UNet conv_out is faster on the generic 64-column tile, the 1 M-pixel VAE conv_out 1.7x faster here).
constexpr long kSmallCoutMinPixels = 131072;
int launch_conv3x3_small_cout(const half_t* x, long ldx, const half_t* w, long K, const float* bias, half_t* y_nchw,
                              int N, int H, int W, int Cin, int Cout, hipStream_t s);
// ---- the two ends of the UNet (edge.hip) ----
// conv_in: x NCHW f16 [N, Cin, H, W] (9 Cin <= 64) -> y NHWC [N H W, 320] (row stride ldy) = conv3x3(x) + bias, packed
// weights [>= 320][64] (k = (kh, kw, ci), zero from 9 Cin up: WeightStore::pack_conv's layout).  gnstat_out != nullptr:
// also the GroupNorm summaries of y for G groups in the IGemmParams::gnstat_out layout with 128-row tiles
// (GnStats{part = gnstat_out, S = H W / 128, rows = 128}).
struct HeadParams {
    const half_t* x_nchw = nullptr;
    const half_t* w = nullptr; long K = 0;
    const float* bias = nullptr;
    half_t* y = nullptr; long ldy = 0;
    float* gnstat_out = nullptr; int G = 0;
    int N = 0, Cin = 0, H = 0, W = 0, Cout = 0;
};
bool conv_head_supported(const HeadParams& p);
int launch_conv_head(const HeadParams& p, hipStream_t s);
// norm_out -> SiLU -> conv_out: y NCHW f16 [N, Cout <= 4, H, W] = conv3x3(act(GroupNorm(x))) + bias; x NHWC (row stride
// ldx), GroupNorm summaries of x as (gn_part, gn_S, gn_rows) (GnStats), packed weights [>= Cout][K = 9 C] in
// pack_conv's K order [C / 64][kh][kw][64].
struct TailParams {
    const half_t* x = nullptr; long ldx = 0;
    const float* gn_part = nullptr; int gn_S = 0; long gn_rows = 0;
    int G = 0; float eps = 0.f; const float* gamma = nullptr; const float* beta = nullptr; int silu = 0;
    const half_t* w = nullptr; long K = 0; const float* bias = nullptr;
    half_t* y = nullptr;
    int N = 0, H = 0, W = 0, C = 0, Cout = 0;
};
bool conv_tail_supported(const TailParams& p);
int launch_conv_tail(const TailParams& p, hipStream_t s);
// Weight packing: OIHW -> [O][KH][KW][I(+pad)] rows; Kpad >= KH*KW*I.
int launch_pack_conv(const half_t* w_oihw, half_t* wp, int O, int I, int KH, int KW, long Kpad,
                     hipStream_t s);
int launch_cfg_duplicate(const half_t* lat, half_t* out, long n_total, float scale, hipStream_t s);
// CLIP text embeddings: out[b, t, :] = token_table[ids[b, t], :] + position_table[t, :]  (ids clamped to the table)
int launch_clip_embed(const int* ids, const half_t* tok, const half_t* pos, half_t* out, int B, int T, int H, int vocab,
                      hipStream_t s);
// out[b, :] = x[b * T + idx[b], :] as f16 (out16) and / or f32 (out32); idx clamped to [0, T)
int launch_gather_rows(const half_t* x, long ldx, const int* idx, half_t* out16, float* out32, int B, int T, int H,
                       hipStream_t s);
int launch_inpaint_blend(half_t* lat, const half_t* img, const half_t* noise, const half_t* mask, float a, float b, int B,
                         int C, long HW, hipStream_t s);
int launch_scale_f16(half_t* x, long n, float scale, hipStream_t s);     // x *= scale (fp32 multiply, one rounding)
int launch_image_to_uint8(const half_t* img, unsigned char* out, int B, int C, long HW, hipStream_t s);
int launch_cfg_linear(const half_t* eps2b, half_t* lat, float* hist, long n, float g, float cx, float ce, float ch,
                      float hx, float he, hipStream_t s);
int launch_cfg_ddim(const half_t* eps2b, half_t* lat, long n, float g, float cx, float ce,
                    hipStream_t s);

// Box probe (probe.hip): back-to-back 16x16x32 f16 MFMA loop (TFLOP/s) and a 16-byte-per-lane copy (GB/s, read + write)
int probe_mfma(int iters, float* tflops, hipStream_t s);
int probe_copy(long bytes, int iters, float* gbs, hipStream_t s);
// L2 -> LDS rate of the LDS-DMA path with every CU streaming (probe.hip); aggregate GB/s over all CUs
int probe_dma(long region_bytes, int passes, int depth, int shared, float* gbs, hipStream_t s);

}  // namespace sd
