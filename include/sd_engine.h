/*
 * sd_engine.h -- C-ABI of the MI355X-native Stable Diffusion denoise engine (libsd_engine.so).
 *
 * The reference (GrafikXxxxxxxYyyyyyyyyyy/StableDiffusion) has no FFI / plugin layer of its own: its
 * hot path is two Python object slots, `SDModelWrapper.base` (UNet2DConditionModel) and
 * `SDModelWrapper.vae` (AutoencoderKL), filled at /root/reference/models/stable_diffusion.py:110-123
 * and called at /root/reference/pipelines/sd_unified_pipeline.py:475-482 (UNet forward) and
 * :523 (VAE decode), :1027-1032 (VAE encode).  The entry points below are what a ctypes binding
 * for those slots binds (INTEGRATION.md shows the stub); each one names the reference interface
 * it replaces.
 *
 * Conventions
 *   - plain C: opaque handles, plain pointers and sizes, int return codes (0 = ok); no torch types.
 *   - every tensor buffer is DEVICE memory owned by the caller; the library owns only its packed
 *     weights and a workspace arena sized on first use of a shape.
 *   - boundary tensors are NCHW fp16, contiguous -- the layout of the reference's tensors; the
 *     engine runs NHWC internally.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and never synchronise.
 *   - errors: non-zero code + thread-local message from sd_last_error(); the Python shim raises
 *     RuntimeError, matching the reference's plain-exception convention
 *     (sd_unified_pipeline.py:302-306).
 *   - single caller thread per handle (the reference's handler is synchronous, rp_handler.py:44-63).
 */
#ifndef SD_ENGINE_H
#define SD_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_OK 0
#define SD_ERR_INVALID 1   /* bad argument / unknown key / shape mismatch */
#define SD_ERR_STATE 2     /* call order violated (e.g. forward before finalize) */
#define SD_ERR_HIP 3       /* HIP runtime error */
#define SD_ERR_UNSUPPORTED 4

#define SD_DTYPE_F16 0
#define SD_DTYPE_F32 1

#define SD_MAX_BLOCKS 4

/* UNet2DConditionModel hyper-parameters; field meaning = diffusers config fields as derived at
 * /root/reference/scripts/convert_from_A1111.py:175-189.  block type 1 = CrossAttn*Block2D, 0 = plain. */
typedef struct sd_unet_config {
    int32_t in_channels;
    int32_t out_channels;
    int32_t num_blocks;
    int32_t block_out_channels[SD_MAX_BLOCKS];
    int32_t down_block_has_attn[SD_MAX_BLOCKS];
    int32_t up_block_has_attn[SD_MAX_BLOCKS];
    int32_t num_heads[SD_MAX_BLOCKS];              /* per down block; reversed for up blocks */
    int32_t transformer_layers[SD_MAX_BLOCKS];     /* per down block; reversed for up blocks */
    int32_t layers_per_block;
    int32_t cross_attention_dim;
    int32_t use_linear_projection;
    int32_t norm_num_groups;
    float   norm_eps;
    int32_t flip_sin_to_cos;
    float   freq_shift;
    int32_t addition_time_embed_dim;               /* 0 = no text_time conditioning (SD1.5) */
    int32_t projection_class_embeddings_input_dim; /* SDXL: 2816 */
} sd_unet_config;

/* AutoencoderKL hyper-parameters (convert_from_A1111.py:490-511). */
typedef struct sd_vae_config {
    int32_t in_channels;      /* 3 */
    int32_t out_channels;     /* 3 */
    int32_t latent_channels;  /* 4 */
    int32_t num_blocks;
    int32_t block_out_channels[SD_MAX_BLOCKS];
    int32_t layers_per_block;
    int32_t norm_num_groups;
} sd_vae_config;

/* CLIP text encoder hyper-parameters (transformers CLIPTextConfig; the reference loads
 * CLIPTextModel / CLIPTextModelWithProjection at models/stable_diffusion.py:124-152). */
typedef struct sd_clip_config {
    int32_t vocab_size;          /* 49408 */
    int32_t hidden_size;         /* 768 (CLIP-L) / 1280 (OpenCLIP bigG) */
    int32_t intermediate_size;   /* 3072 / 5120 */
    int32_t num_layers;          /* 12 / 32 */
    int32_t num_heads;           /* 12 / 20 */
    int32_t max_positions;       /* 77 */
    int32_t hidden_act;          /* 0 quick_gelu, 1 gelu (erf) */
    int32_t projection_dim;      /* 0: CLIPTextModel; > 0: CLIPTextModelWithProjection.text_projection */
    float layer_norm_eps;        /* 1e-5 */
} sd_clip_config;

typedef struct sd_unet sd_unet;
typedef struct sd_vae sd_vae;
typedef struct sd_clip sd_clip;

/* -- library ------------------------------------------------------------------------------- */
const char* sd_last_error(void);
int sd_engine_version(void);
/* Name of the code object's ISA target ("gfx950"). */
const char* sd_engine_arch(void);

/* -- UNet: replaces the object in SDModelWrapper.base (stable_diffusion.py:117-123) ---------- */
int sd_unet_create(const sd_unet_config* cfg, sd_unet** out);
int sd_unet_destroy(sd_unet* u);
/* Number of weight tensors the model expects / name and rank+shape of the i-th one, in
 * diffusers state-dict naming (convert_from_A1111.py:240-485). */
int sd_unet_num_weights(const sd_unet* u);
int sd_unet_weight_info(const sd_unet* u, int index, const char** key, int64_t* shape4, int* ndim);
/* Hand one tensor over (host or device pointer, contiguous, PyTorch layout:
 * conv [Cout,Cin,KH,KW], linear [out,in], vectors [C]).  The library copies + repacks; the
 * caller's buffer can be freed on return.  Replaces state_dict loading done by
 * UNet2DConditionModel.from_pretrained (stable_diffusion.py:117-123). */
int sd_unet_set_weight(sd_unet* u, const char* diffusers_key, const void* data,
                       const int64_t* shape, int ndim, int dtype);
/* All weights present -> pre-pack (fused qkv, GEGLU interleave, time-emb projection stack). */
int sd_unet_finalize(sd_unet* u);
/* UNet2DConditionModel.forward as called at sd_unified_pipeline.py:475-482.
 *   sample     [B,Cin,H,W] f16      latent_model_input
 *   timesteps  [B] f32 (device)     `t` broadcast to the batch
 *   ehs        [B,L,D] f16          prompt_embeds
 *   add_text   [B,P] f16 or NULL    added_cond_kwargs["text_embeds"]  (SDXL, :430-433)
 *   add_time_ids [B,6] f32 or NULL  added_cond_kwargs["time_ids"]
 *   out        [B,Cout,H,W] f16     noise_pred
 */
int sd_unet_forward(sd_unet* u, const void* sample, const float* timesteps, const void* ehs,
                    int ehs_len, const void* add_text, const float* add_time_ids, void* out,
                    int B, int H, int W, void* stream);
/* Replay the whole forward from a captured hipGraph (one per input shape; inputs / output staged
 * through engine-owned buffers, fenced against `stream` with events).  Host cost per forward drops
 * from ~480 kernel launches to one hipGraphLaunch; results are bitwise those of the eager path. */
int sd_unet_use_graph(sd_unet* u, int enable);
/* Text K/V reuse across the forwards of ONE denoise loop (sd_unified_pipeline.py:465-507 passes the same
 * prompt_embeds to every step): enable = 1 makes the next forward compute the stacked attn2.to_k / to_v
 * projections of encoder_hidden_states and later forwards with the same (pointer, batch, length) reuse
 * them.  Every call of this function -- with 1 or 0 -- invalidates what is cached: call it (again) whenever
 * the CONTENTS behind the pointer may have changed, i.e. at the start of each pipeline call.  Off by default. */
int sd_unet_text_kv_cache(sd_unet* u, int enable);
/* Bytes of device memory held (packed weights, workspace). */
int sd_unet_memory(const sd_unet* u, int64_t* weight_bytes, int64_t* workspace_bytes);

/* -- VAE: replaces the object in SDModelWrapper.vae (stable_diffusion.py:110-116) ------------ */
int sd_vae_create(const sd_vae_config* cfg, sd_vae** out);
int sd_vae_destroy(sd_vae* v);
int sd_vae_num_weights(const sd_vae* v);
int sd_vae_weight_info(const sd_vae* v, int index, const char** key, int64_t* shape4, int* ndim);
int sd_vae_set_weight(sd_vae* v, const char* diffusers_key, const void* data,
                      const int64_t* shape, int ndim, int dtype);
int sd_vae_finalize(sd_vae* v);
/* AutoencoderKL.decode(z)[0] (sd_unified_pipeline.py:523): z [B,4,h,w] f16 -> img [B,3,8h,8w] f16. */
int sd_vae_decode(sd_vae* v, const void* z, void* img, int B, int h, int w, void* stream);
/* AutoencoderKL.encode(x) up to the moments (sd_unified_pipeline.py:1027-1032):
 * img [B,3,H,W] f16 -> moments [B,8,H/8,W/8] f16 (mean | logvar); sampling stays host code. */
int sd_vae_encode(sd_vae* v, const void* img, void* moments, int B, int H, int W, void* stream);
/* `vae.config.force_upcast` (sd_unified_pipeline.py:1020-1036): the reference runs such a VAE (SDXL's) in float32 around
 * encode because its activations leave fp16's range.  The engine instead stores every inter-layer activation of the encoder
 * 2^-shift times smaller -- GroupNorm is invariant to the scale of its input (eps is scaled along), so the encoder computes
 * the same function with fp32 accumulators and statistics as before; shift = 0 (default) is plain fp16 storage.  Applies
 * to the following sd_vae_encode calls of this handle. */
int sd_vae_encode_range_shift(sd_vae* v, int shift);
int sd_vae_memory(const sd_vae* v, int64_t* weight_bytes, int64_t* workspace_bytes);

/* -- CLIP text encoder: replaces SDModelWrapper.text_encoder / .text_encoder_2 as encode_prompt calls
 *    them (sd_unified_pipeline.py:592-608).  Weight names are the
 *    transformers state-dict keys ("text_model.embeddings.token_embedding.weight", ...,
 *    "text_projection.weight"). ------------------------------------------------------------------ */
int sd_clip_create(const sd_clip_config* cfg, sd_clip** out);
int sd_clip_destroy(sd_clip* c);
int sd_clip_num_weights(const sd_clip* c);
int sd_clip_weight_info(const sd_clip* c, int index, const char** key, int64_t* shape4, int* ndim);
int sd_clip_set_weight(sd_clip* c, const char* key, const void* data, const int64_t* shape, int ndim, int dtype);
int sd_clip_finalize(sd_clip* c);
/* text_encoder(input_ids, output_hidden_states=True): ids int32 [B,T] (device).  Outputs, each
 * nullable, all f16 on the device:
 *   hidden_states [num_layers+1, B, T, H]  (embeddings, then every layer's output; no final norm)
 *   last_hidden   [B, T, H]                (final_layer_norm of the last one)
 *   pooled        [B, H]                   (last_hidden at eos_index[b]; pooler_output)
 *   text_embeds   [B, projection_dim]      (text_projection(pooled); needs projection_dim > 0)
 * eos_index int32 [B] (device) is required for pooled / text_embeds: the host picks it the way
 * transformers does (argmax of the ids, or first eos_token_id). */
int sd_clip_forward(sd_clip* c, const int32_t* input_ids, const int32_t* eos_index, void* hidden_states,
                    void* last_hidden, void* pooled, void* text_embeds, int B, int T, void* stream);
/* text_encoder.text_model.final_layer_norm(x) for the clip_skip branch (sd_unified_pipeline.py:608). */
int sd_clip_final_layer_norm(sd_clip* c, const void* x, void* y, int64_t rows, void* stream);
int sd_clip_memory(const sd_clip* c, int64_t* weight_bytes, int64_t* workspace_bytes);

/* -- denoise-step glue (sd_unified_pipeline.py:467-469, :484-489) ---------------------------- */
/* latent_model_input = cat([latents]*2) * in_scale   (in_scale = 1 for DDIM / DPM++) */
int sd_cfg_duplicate(const void* latents, void* out2b, int64_t n_per_batch, int B, float in_scale,
                     void* stream);
/* noise = u + g (t - u);  x <- c_x * x + c_eps * noise   (DDIM eta=0 written as an affine update;
 * coefficients computed on the host by the scheduler).  noise_pred_2b = [uncond ; text]. */
int sd_cfg_ddim_step(const void* noise_pred_2b, void* latents, int64_t n, float guidance_scale,
                     float c_x, float c_eps, void* stream);

/* The same for every scheduler of the reference's registry (stable_diffusion.py:199-227) whose update
 * is linear in (x, eps, previous x0 prediction) -- DDIM, Euler, DPM-Solver++(2M):
 *   eps = u + g (t - u);  x0 = h_x x + h_eps eps;  x <- c_x x + c_eps eps + c_hist hist;  hist <- x0
 * hist_f32 [n] is the scheduler's history (nullable: then c_hist / h_* are ignored).  Coefficients
 * come from the host scheduler (schedulers.py `fused_plan`), replacing scheduler.step at :489. */
int sd_cfg_linear_step(const void* noise_pred_2b, void* latents, float* hist_f32, int64_t n,
                       float guidance_scale, float c_x, float c_eps, float c_hist, float h_x, float h_eps,
                       void* stream);

/* Inpainting with a 4-channel UNet, after every scheduler step (sd_unified_pipeline.py:492-506):
 *   latents <- m latents + (1 - m) (a image_latents + b noise),  m = mask [B,1,H,W] f16 over channels;
 * (a, b) = scheduler.add_noise coefficients at the NEXT timestep, or noise = NULL on the last step. */
int sd_inpaint_blend(void* latents, const void* image_latents, const void* noise, const void* mask,
                     float a, float b, int B, int C, int H, int W, void* stream);

/* convert_pt_to_numpy (runpod-worker/handler_logic.py:21-29): decoded images [B,C,H,W] f16 in [-1,1] ->
 * [B,H,W,C] uint8, with the reference's fp16 roundings and truncating cast (bit-exact with running
 * the reference's op sequence on the same fp16 tensor).  C <= 4. */
int sd_images_to_uint8(const void* images_nchw_f16, void* out_nhwc_u8, int B, int C, int H, int W, void* stream);

/* -- per-kernel timing for bench.py's live roofline ----------------------------------------- */
/* While enabled, every conv / norm / attention launch of the models is bracketed by HIP events on
 * the launch stream.  sd_prof_collect synchronises and returns one aggregate per kernel name:
 * algorithmic FLOPs and bytes (2*MAC; inputs + weights + outputs once), summed event time. */
typedef struct sd_prof_entry {
    char kernel[64];
    double flops;
    double bytes;
    double ms;
    int64_t launches;
} sd_prof_entry;
int sd_prof_enable(int on);
int sd_prof_collect(sd_prof_entry* out, int max_entries, int* n_entries);

/* Box probe for bench.py (`box_probe`): model-independent microbenchmarks run in-process next to the timed
 * region so that `value` can be read against the box it ran on (the reference has no counterpart; it serves the
 * measurement contract only).  sd_probe_mfma: `iters` rounds of 16 back-to-back v_mfma_f32_16x16x32_f16 per wave,
 * one wave per SIMD on every CU, random operands -> dense fp16 TFLOP/s.  sd_probe_copy: `iters` passes of a
 * 16-byte-per-lane copy of `bytes` (choose > 256 MiB to pass the Infinity Cache) -> GB/s, read + write. */
int sd_probe_mfma(int iters, float* tflops, void* stream);
int sd_probe_copy(int64_t bytes, int iters, float* gbs, void* stream);
/* L2 -> LDS rate of the LDS-DMA path (buffer_load ... lds) with every CU streaming, the operand path of the GEMM kernels:
 * each block (one per CU, four waves) walks `region_bytes` (a multiple of 4096) `passes` times with `depth` 1-KiB pieces
 * outstanding per wave (1, 2, 4, 8, 16 or 32); shared bit 0: all blocks walk the same region (a weight operand), else each its
 * own (an activation operand); shared bit 1: the same walk with ordinary 16-byte loads into registers.  *gbs = aggregate GB/s.  DESIGN.md section 4 reads the GEMM family's ceiling against it. */
int sd_probe_lds_dma(int64_t region_bytes, int passes, int depth, int shared, float* gbs, void* stream);

/* Tuner / test hook: force the LDS-DMA conv kernel's tile variant (0..5) and split-K factor for
 * every following launch; variant -1 restores the built-in per-shape choice. */
int sd_igemm_force(int variant, int splits);

/* -- single operators, exported for the parity tests (tests/test_ops_gpu.py) ----------------- */
/* Implicit-GEMM convolution / linear on NHWC f16:
 *   y[n,oh,ow,co] = bias[co] + rowadd[n,co] + res[n,oh,ow,co]
 *                 + sum_{kh,kw,ci} x[n, (oh*stride+kh-pad)>>up, (ow*stride+kw-pad)>>up, ci] * w[co,kh,kw,ci]
 * w is PyTorch [Cout,Cin,KH,KW] (f16, device), bias [Cout] / rowadd [N,Cout] f32 device (nullable);
 * packed internally per call (test path only; synchronises). */
int sd_op_conv2d(const void* x_nhwc, const void* w_oihw, const void* bias, const void* rowadd,
                 const void* res_nhwc, void* y_nhwc, int N, int H, int W, int Cin, int Cout,
                 int ksize, int stride, int upsample2x, int geglu, void* stream);
/* conv_out: 3x3 / stride 1 / pad 1 convolution to 1..4 output channels (UNet2DConditionModel.conv_out 320 -> 4,
 * AutoencoderKL decoder.conv_out 128 -> 3; diffusers modules under sd_unified_pipeline.py:475-482, :523) with
 * the NHWC -> NCHW change of layout fused: x NHWC f16, w OIHW f16, bias f32, y NCHW f16.  Cin % 64 == 0. */
int sd_op_conv3x3_small_cout(const void* x_nhwc, const void* w_oihw, const void* bias, void* y_nchw, int N, int H,
                             int W, int Cin, int Cout, void* stream);
/* Convolution followed by GroupNorm (+ SiLU) of its output, the pair ResnetBlock2D issues as
 * conv1 -> norm2 and the VAE decoder as conv2 -> next norm1 (diffusers resnet.py under
 * sd_unified_pipeline.py:475-482, :523).  When the launch allows it the convolution's epilogue leaves
 * the GroupNorm statistics and the GroupNorm makes no pass of its own over y_conv;
 * *stats_from_epilogue (may be NULL) tells which path ran.  y_conv and y_gn are both written. */
int sd_op_conv2d_groupnorm(const void* x_nhwc, const void* w_oihw, const void* bias, const void* rowadd,
                           const void* res_nhwc, void* y_conv_nhwc, const void* gamma, const void* beta,
                           void* y_gn_nhwc, int N, int H, int W, int Cin, int Cout, int ksize, int stride,
                           int upsample2x, int groups, float eps, int silu, int* stats_from_epilogue, void* stream);
/* GroupNorm (+ SiLU) followed by a convolution, the pair ResnetBlock2D issues twice (norm1 -> SiLU -> conv1,
 * norm2 -> SiLU -> conv2; diffusers resnet.py under sd_unified_pipeline.py:475-482, :523):
 *   y = conv(act(GroupNorm(x; gamma, beta, groups, eps))) + bias + rowadd + res
 * For 3x3 / stride-1 convolutions the convolution applies the norm to its input tiles in LDS and the normalised
 * tensor never exists in HBM (*fused = 1); otherwise a GroupNorm kernel runs first (*fused = 0).  iters > 0: timed
 * like sd_bench_conv2d (the statistics pass is outside the timed launches), ms_per_launch written. */
int sd_op_groupnorm_conv2d(const void* x_nhwc, const void* gamma, const void* beta, int groups, float eps, int silu,
                           const void* w_oihw, const void* bias, const void* rowadd, const void* res_nhwc, void* y_nhwc,
                           int N, int H, int W, int Cin, int Cout, int ksize, int iters, float* ms_per_launch, int* fused,
                           void* stream);
/* The two ends of the UNet as one launch each (edge.hip; diffusers UNet2DConditionModel.forward: conv_in, and
 * conv_norm_out -> conv_act -> conv_out, under sd_unified_pipeline.py:475-482).
 * sd_op_unet_conv_in: y_nhwc [N H W, 320] f16 = conv3x3(x_nchw [N, Cin, H, W] f16; w_oihw [320, Cin, 3, 3] f16) + bias (f32),
 *   9 Cin <= 64 and H W % 128 == 0.  groups > 0: gn_summaries [N][H W / 128][groups][2] f32 (device) receives (mean, M2) of
 *   every 128-pixel tile x group of the stored output -- what the first resnet's GroupNorm merges instead of reading y.
 * sd_op_unet_conv_out: y_nchw [N, Cout <= 4, H, W] f16 = conv3x3(act(GroupNorm(x_nhwc [N H W, 320]; gamma, beta, groups, eps));
 *   w_oihw [Cout, 320, 3, 3]) + bias, act = SiLU when silu != 0; H % 8 == 0, W % 16 == 0, groups <= 32.
 * Both return SD_ERR_INVALID for shapes the one-launch kernels do not take (the UNet then runs its general path);
 * iters > 0: timed like sd_bench_conv2d (packing and the statistics pass outside the timed launches). */
int sd_op_unet_conv_in(const void* x_nchw, const void* w_oihw, const void* bias, void* y_nhwc, float* gn_summaries, int groups,
                       int N, int Cin, int H, int W, int Cout, int iters, float* ms_per_launch, void* stream);
int sd_op_unet_conv_out(const void* x_nhwc, const void* gamma, const void* beta, int groups, float eps, int silu,
                        const void* w_oihw, const void* bias, void* y_nchw, int N, int H, int W, int C, int Cout, int iters,
                        float* ms_per_launch, void* stream);
/* The GEGLU feed-forward of BasicTransformerBlock with its norm and residual (diffusers attention.py: norm3 -> FeedForward
 * (GEGLU) -> + hidden_states, under sd_unified_pipeline.py:475-482):
 *   y = x + (h * gelu(g)) W2^T + b2,   [h | g] = LayerNorm(x; gamma, beta, eps) W1^T + b1
 * x, y [M, C] f16; w1 [8C, C], w2 [C, 4C] f16 (PyTorch Linear layout); gamma, beta, b1 [8C], b2 [C] f32.  For C = 320 and
 * M % 128 == 0 one launch keeps the 4C-wide hidden tensor on the CU (*fused = 1, ffn.hip); otherwise the projection with
 * its GEGLU epilogue and the output linear with its residual epilogue run (*fused = 0).  iters > 0: ms_per_launch[0] = the
 * path taken, ms_per_launch[1] = the two-GEMM form on the same operands (packing outside the timed launches). */
int sd_op_ffn_geglu(const void* x, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* w1, const void* b1,
                    const void* w2, const void* b2, void* y, int M, int C, int iters, float* ms_per_launch, int* fused,
                    void* stream);
/* Same operator, timed: `iters` back-to-back launches bracketed by HIP events on `stream`
 * (after two warm-up launches); used by tools/tune_igemm.py to pick tile variants per shape. */
int sd_bench_conv2d(const void* x_nhwc, const void* w_oihw, void* y_nhwc, int N, int H, int W, int Cin,
                    int Cout, int ksize, int stride, int upsample2x, int geglu, int iters,
                    float* ms_per_launch, void* stream);
/* GroupNorm (+ optional SiLU) on NHWC f16, fp32 statistics. */
int sd_op_groupnorm(const void* x_nhwc, const void* gamma, const void* beta, void* y_nhwc,
                    int N, int HW, int C, int groups, float eps, int silu, void* stream);
/* GroupNorm of a channel concatenation [A | B] (diffusers' up blocks: torch.cat([hidden_states, res_hidden_states], 1) ->
 * ResnetBlock2D.norm1, under sd_unified_pipeline.py:475-482) from per-half summaries, the way the UNet runs it on its big
 * maps: statistics of the first Ca channels over sub-groups of width gcd(C / groups, Ca), of the last Cb over their own
 * `groups` groups, one small launch merging them per group of the concatenation, then the apply pass.  x, y [N HW, Ca + Cb]
 * f16.  SD_ERR_INVALID when the halves' sub-groups cannot tile the groups (the UNet then runs an ordinary statistics pass). */
int sd_op_groupnorm_concat(const void* x_nhwc, int Ca, int Cb, const void* gamma, const void* beta, void* y_nhwc, int N, int HW,
                           int groups, float eps, int silu, void* stream);
/* Same operator, timed like sd_bench_conv2d (scratch allocated once, `iters` launches between HIP events). */
int sd_bench_groupnorm(const void* x_nhwc, const void* gamma, const void* beta, void* y_nhwc,
                       int N, int HW, int C, int groups, float eps, int silu, int iters,
                       float* ms_per_launch, void* stream);
/* Timesteps / get_timestep_embedding as UNet2DConditionModel.time_proj and SDXL's add_time_proj issue it
 * (diffusers embeddings.py under sd_unified_pipeline.py:475-482): out[b, :] = [cos(t_b f_i) | sin(t_b f_i)]
 * (flip_sin_to_cos = 1) or [sin | cos], f_i = exp(-ln(1e4) i / (dim/2 - freq_shift)).  t, out: f32 device. */
int sd_op_timestep_sinusoid(const float* t, float* out, int count, int dim, int flip_sin_to_cos, float freq_shift,
                            void* stream);
/* The small-batch linear of the time-embedding MLPs (TimestepEmbedding.linear_1 / linear_2, time_emb_proj):
 * y[b, n] = act_out(bias[n] + sum_k act_in(x[b, k]) * W[n, k]), act = SiLU when the flag is set.  x, bias, y f32,
 * W f16 row-major [n_out, k]. */
int sd_op_small_linear(const float* x, const void* w_f16, const float* bias, float* y, int B, int K, int n_out,
                       int silu_in, int silu_out, void* stream);
/* LayerNorm over the last dim of [rows, C] f16. */
int sd_op_layernorm(const void* x, const void* gamma, const void* beta, void* y, int rows, int C,
                    float eps, void* stream);
/* softmax(q k^T / sqrt(d)) v.  q [B,Tq,heads*d] (row stride ldq), k/v [B,Tk,heads*d], out like q. */
int sd_op_attention(const void* q, const void* k, const void* v, void* out, int B, int Tq, int Tk,
                    int heads, int d, int ldq, int ldk, int ldv, int ldo, void* stream);
/* The same with options.  causal = 1: key j > query i contributes nothing (CLIP text self-attention).
 * prescaled = 1: q already carries log2(e)/sqrt(d) (the UNet folds it into its query projections),
 * the kernel then skips its per-score scaling. */
int sd_op_attention_ex(const void* q, const void* k, const void* v, void* out, int B, int Tq, int Tk,
                       int heads, int d, int ldq, int ldk, int ldv, int ldo, int causal, int prescaled,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SD_ENGINE_H */
