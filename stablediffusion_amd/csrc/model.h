// Model graphs (UNet2DConditionModel, AutoencoderKL) assembled from the gfx950 kernels.
#pragma once
#include "../../include/sd_engine.h"
#include "engine.h"

namespace sd {

const std::string& last_error();

struct Resnet {
    NormW n1, n2;
    ConvW c1, c2, sc;
    bool has_sc = false;
    int temb_off = -1;
    int cin = 0, cout = 0;
};
struct TBlock {
    NormW ln1, ln2, ln3;
    ConvW qkv, out1, q2, out2, ff1, ff2;
    int kv_off = 0;     // column offset of this block's [K | V] text projection in UNet::kv_all
    bool fold = false;  // ln1 / ln2 / ln3 are folded into qkv / q2 / ff1 (no LayerNorm launches)
};
struct Xformer {
    NormW gn;
    ConvW pin, pout;
    std::vector<TBlock> blocks;
    int C = 0, heads = 1;
};
struct VaeAttn {
    NormW gn;
    ConvW qkv, out;
    int C = 0;
};

// x_stats: GroupNorm summaries of x left by the convolution that produced it (or nullptr);
// *out_stats: where the block leaves the summaries of `out` for the GroupNorm that consumes it next.
// stream_scale s (a power of two, the VAE encoder's range-scaled form): x and out hold s times the block's true input /
// output.  GroupNorm is invariant to the scale of its input up to eps, so norm1 / norm2 run with eps * s^2, conv1 /
// conv2 emit s * (conv + bias), the shortcut (whose input carries s already) only scales its bias: the block computes the
// same function, with every stored activation s times smaller.
// out_buf / out_groups: where and at which granularity those summaries are written (default: the next ring slot, G
// groups) -- a skip connection's live until the up path reads them, and an output that becomes the hidden half of a
// concatenation is summarised over the sub-groups that concatenation's GroupNorm can merge (gn_cat_unit).
void run_resnet(Ctx& c, const Resnet& r, View x, int N, int H, int W, View out, int G, float eps,
                const float* tproj, int tproj_ld, const GnStatBuf* x_stats = nullptr, GnStatBuf** out_stats = nullptr,
                float stream_scale = 1.f, GnStatBuf* out_buf = nullptr, int out_groups = 0);
void run_xformer(Ctx& c, const Xformer& t, View x, int N, int H, int W, View out, int G, View text_kv, int L,
                 const GnStatBuf* x_stats = nullptr, GnStatBuf** out_stats = nullptr, GnStatBuf* out_buf = nullptr,
                 int out_groups = 0);

struct UNet {
    explicit UNet(const sd_unet_config& c);
    int finalize();
    int forward(const half_t* sample, const float* timesteps, const half_t* ehs, int L,
                const half_t* add_text, const float* add_time_ids, half_t* out, int B, int H, int W,
                hipStream_t stream);

    sd_unet_config cfg;
    WeightStore ws;
    Arena arena;
    bool finalized = false;
    long planned_key = -1;

    // ---- optional hipGraph replay of the whole forward (sd_unet_use_graph) ----
    // One captured graph per input shape; inputs / output are staged through engine-owned buffers so
    // the captured pointers stay valid whatever tensors the caller passes.  The ~480 launches of a
    // forward then cost the host one hipGraphLaunch (CPU time per step 2 ms -> ~0.02 ms); GPU time is
    // unchanged (kernel boundaries cost the same under a graph, MI355X_MICROARCH.md price list).
    bool graph_enabled = false;
    hipStream_t gstream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraphExec_t gexec = nullptr;
    long graph_key = -1;
    char* io_slab = nullptr;
    size_t io_cap = 0;
    int forward_graph(const half_t* sample, const float* timesteps, const half_t* ehs, int L, const half_t* add_text,
                      const float* add_time_ids, half_t* out, int B, int H, int W, hipStream_t stream);
    ~UNet();

    // ---- text K/V kept across the forwards of one denoise loop (sd_unet_text_kv_cache) ----
    // encoder_hidden_states is constant over the loop, so the stacked to_k / to_v projection of every
    // cross-attention block is computed by the first forward and reused by the other 49.
    // Valid for one (pointer, batch, length); the caller invalidates when the CONTENTS behind the pointer change.
    bool kv_cache_on = false, kv_valid = false;
    half_t* kv_cache = nullptr;
    size_t kv_cap = 0;
    const half_t* kv_src = nullptr;
    int kv_B = 0, kv_L = 0;

    ConvW conv_in, conv_out, te1, te2, ae1, ae2, temb_stack;
    ConvW kv_all;                       // every attn2.to_k / to_v of the model, row-concatenated
    std::vector<std::string> kv_keys;   // (finalize only)
    NormW norm_out;
    std::vector<std::vector<Resnet>> down_res, up_res;
    std::vector<std::vector<Xformer>> down_att, up_att;
    std::vector<ConvW> down_ds, up_us;
    Resnet mid_r0, mid_r1;
    Xformer mid_att;
    int temb_total = 0;
    int kv_total = 0;

  private:
    int pack_resnet(const std::string& p, Resnet* r, std::vector<std::string>* tw, std::vector<std::string>* tb);
    int pack_xformer(const std::string& p, Xformer* x, int heads, int depth);
    int run(Ctx& c, const half_t* sample, const float* timesteps, const half_t* ehs, int L,
            const half_t* add_text, const float* add_time_ids, half_t* out, int B, int H, int W);
};

struct VAE {
    explicit VAE(const sd_vae_config& c);
    int finalize();
    int decode(const half_t* z, half_t* img, int B, int h, int w, hipStream_t stream);
    int encode(const half_t* img, half_t* moments, int B, int H, int W, hipStream_t stream);
    // The encoder with every inter-layer activation stored 2^-k times smaller (same function: see run_resnet): the
    // engine's answer to `config.force_upcast` (sd_unified_pipeline.py:1020-1036 runs such VAEs in fp32 around encode
    // because their activations leave fp16's range).  0 = plain fp16 storage.
    int encode_shift = 0;

    sd_vae_config cfg;
    WeightStore ws;
    Arena arena;
    bool finalized = false;
    long planned_key = -1;

    // decoder
    half_t* pq_w = nullptr; float* pq_b = nullptr;      // post_quant_conv (pointwise, NCHW)
    ConvW d_conv_in, d_conv_out;
    NormW d_norm_out;
    Resnet d_mid0, d_mid1;
    VaeAttn d_attn;
    std::vector<std::vector<Resnet>> d_up;
    std::vector<ConvW> d_us;
    // encoder
    half_t* q_w = nullptr; float* q_b = nullptr;        // quant_conv
    ConvW e_conv_in, e_conv_out;
    NormW e_norm_out;
    Resnet e_mid0, e_mid1;
    VaeAttn e_attn;
    std::vector<std::vector<Resnet>> e_down;
    std::vector<ConvW> e_ds;

  private:
    int pack_resnet(const std::string& p, Resnet* r);
    int pack_attn(const std::string& p, VaeAttn* a);
    int pack_pointwise(const std::string& p, half_t** w, float** b);
    void run_attn(Ctx& c, const VaeAttn& a, View x, int N, int H, int W, View out, const GnStatBuf* x_stats,
                  GnStatBuf** out_stats, float stream_scale = 1.f);
    int run_decode(Ctx& c, const half_t* z, half_t* img, int B, int h, int w);
    int run_encode(Ctx& c, const half_t* img, half_t* moments, int B, int H, int W, float stream_scale);
};

// CLIP text encoder (transformers CLIPTextModel / CLIPTextModelWithProjection): the text side of
// encode_prompt, /root/reference/pipelines/sd_unified_pipeline.py:583-608.
struct ClipLayer {
    NormW ln1, ln2;
    ConvW qkv, out, fc1, fc2;
};
struct CLIP {
    explicit CLIP(const sd_clip_config& c);
    int finalize();
    int forward(const int* ids, const int* eos_index, half_t* hidden_states, half_t* last_hidden, half_t* pooled,
                half_t* text_embeds, int B, int T, hipStream_t stream);
    int final_layer_norm(const half_t* x, half_t* y, long rows, hipStream_t stream);

    sd_clip_config cfg;
    WeightStore ws;
    Arena arena;
    bool finalized = false;
    long planned_key = -1;
    half_t* tok = nullptr;      // [vocab][H]
    half_t* pos = nullptr;      // [max_positions][H]
    half_t* proj = nullptr;     // [projection_dim][H], no bias
    std::vector<ClipLayer> layers;
    NormW final_ln;

  private:
    int run(Ctx& c, const int* ids, const int* eos_index, half_t* hidden_states, half_t* last_hidden, half_t* pooled,
            half_t* text_embeds, int B, int T);
};

}  // namespace sd
