// CLIP text encoder graph (transformers CLIPTextModel / CLIPTextModelWithProjection) on the gfx950
// kernels: what encode_prompt calls at /root/reference/pipelines/sd_unified_pipeline.py:592-608.
// Pre-LN transformer, causal self-attention over 77 tokens, quick_gelu (CLIP-L) or erf-gelu
// (OpenCLIP bigG) MLP.  Per layer: LN -> fused q|k|v GEMM -> causal flash attention -> out-proj GEMM
// (+residual) -> LN -> fc1 GEMM (+activation in the epilogue) -> fc2 GEMM (+residual).  Each layer
// writes its output straight into its slab of the hidden_states output, so "output_hidden_states"
// costs no copies.
#include "model.h"

namespace sd {

namespace {
std::string lkey(int i, const char* rest) { return "text_model.encoder.layers." + std::to_string(i) + "." + rest; }
}  // namespace

CLIP::CLIP(const sd_clip_config& c) : cfg(c) {
    const int H = cfg.hidden_size, I = cfg.intermediate_size;
    ws.declare("text_model.embeddings.token_embedding.weight", {cfg.vocab_size, H});
    ws.declare("text_model.embeddings.position_embedding.weight", {cfg.max_positions, H});
    for (int i = 0; i < cfg.num_layers; ++i) {
        for (const char* n : {"self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.out_proj"}) {
            ws.declare(lkey(i, n) + ".weight", {H, H});
            ws.declare(lkey(i, n) + ".bias", {H});
        }
        ws.declare(lkey(i, "layer_norm1.weight"), {H});
        ws.declare(lkey(i, "layer_norm1.bias"), {H});
        ws.declare(lkey(i, "mlp.fc1.weight"), {I, H});
        ws.declare(lkey(i, "mlp.fc1.bias"), {I});
        ws.declare(lkey(i, "mlp.fc2.weight"), {H, I});
        ws.declare(lkey(i, "mlp.fc2.bias"), {H});
        ws.declare(lkey(i, "layer_norm2.weight"), {H});
        ws.declare(lkey(i, "layer_norm2.bias"), {H});
    }
    ws.declare("text_model.final_layer_norm.weight", {H});
    ws.declare("text_model.final_layer_norm.bias", {H});
    if (cfg.projection_dim > 0) ws.declare("text_projection.weight", {cfg.projection_dim, H});
}

int CLIP::finalize() {
    if (finalized) return 0;
    std::string missing;
    if (!ws.complete(&missing)) { set_error("finalize: weight not set: " + missing); return 2; }
    auto keep = [&](const std::string& key, half_t** dst) -> int {
        const RawTensor* r = ws.raw(key);
        if (!r) { set_error("missing weight: " + key); return 2; }
        *dst = static_cast<half_t*>(ws.dmalloc((size_t)r->numel * sizeof(half_t)));
        if (!*dst) { set_error("hipMalloc failed"); return 3; }
        SD_HIP_CHECK(hipMemcpy(*dst, r->dev, (size_t)r->numel * sizeof(half_t), hipMemcpyDeviceToDevice));
        return 0;
    };
    int rc;
    if ((rc = keep("text_model.embeddings.token_embedding.weight", &tok))) return rc;
    if ((rc = keep("text_model.embeddings.position_embedding.weight", &pos))) return rc;
    if (cfg.projection_dim > 0 && (rc = keep("text_projection.weight", &proj))) return rc;
    layers.assign((size_t)cfg.num_layers, ClipLayer());
    for (int i = 0; i < cfg.num_layers; ++i) {
        ClipLayer& l = layers[(size_t)i];
        if ((rc = ws.pack_norm(lkey(i, "layer_norm1"), &l.ln1))) return rc;
        if ((rc = ws.pack_rows({lkey(i, "self_attn.q_proj.weight"), lkey(i, "self_attn.k_proj.weight"),
                                lkey(i, "self_attn.v_proj.weight")},
                               {lkey(i, "self_attn.q_proj.bias"), lkey(i, "self_attn.k_proj.bias"),
                                lkey(i, "self_attn.v_proj.bias")}, &l.qkv))) return rc;
        if ((rc = ws.pack_conv(lkey(i, "self_attn.out_proj"), &l.out))) return rc;
        if ((rc = ws.pack_norm(lkey(i, "layer_norm2"), &l.ln2))) return rc;
        if ((rc = ws.pack_conv(lkey(i, "mlp.fc1"), &l.fc1))) return rc;
        if ((rc = ws.pack_conv(lkey(i, "mlp.fc2"), &l.fc2))) return rc;
    }
    if ((rc = ws.pack_norm("text_model.final_layer_norm", &final_ln))) return rc;
    SD_HIP_CHECK(hipDeviceSynchronize());
    ws.free_raw();
    finalized = true;
    return 0;
}

int CLIP::run(Ctx& c, const int* ids, const int* eos_index, half_t* hidden_states, half_t* last_hidden, half_t* pooled,
              half_t* text_embeds, int B, int T) {
    Arena& a = *c.arena;
    const int H = cfg.hidden_size, I = cfg.intermediate_size, L = cfg.num_layers;
    const int d = H / cfg.num_heads;
    const long M = (long)B * T;
    const float eps = cfg.layer_norm_eps;
    const int act = cfg.hidden_act == 0 ? 1 : 2;
    hipStream_t s = c.stream;
    const bool go = !c.dry;

    // residual stream: the caller's hidden_states slabs, or two ping-pong buffers
    half_t* pp[2] = {nullptr, nullptr};
    if (!hidden_states) { pp[0] = a.alloc_h(M * H); pp[1] = a.alloc_h(M * H); }
    auto slab = [&](int l) { return hidden_states ? hidden_states + (long)l * M * H : pp[l & 1]; };

    if (go && !c.err) c.err = launch_clip_embed(ids, tok, pos, slab(0), B, T, H, cfg.vocab_size, s);
    View n(a.alloc_h(M * H), H, H), qkv(a.alloc_h(M * 3 * H), 3 * H, 3 * H), att(a.alloc_h(M * H), H, H);
    View t2(a.alloc_h(M * H), H, H), f(a.alloc_h(M * I), I, I);
    for (int l = 0; l < L; ++l) {
        const ClipLayer& w = layers[(size_t)l];
        View x(slab(l), H, H), y(slab(l + 1), H, H);
        op_layernorm(c, w.ln1, x, n, M, eps);
        op_conv(c, w.qkv, n, B, T, 1, qkv);
        op_attention(c, qkv.slice(0, H), qkv.slice(H, H), qkv.slice(2 * H, H), att, B, T, T, cfg.num_heads, d, 1);
        op_conv(c, w.out, att, B, T, 1, t2, 1, 0, nullptr, 0, &x);
        op_layernorm(c, w.ln2, t2, n, M, eps);
        op_conv(c, w.fc1, n, B, T, 1, f, 1, 0, nullptr, 0, nullptr, 0, -1, act);
        op_conv(c, w.fc2, f, B, T, 1, y, 1, 0, nullptr, 0, &t2);
    }
    if (last_hidden || pooled || text_embeds) {
        half_t* lh = last_hidden ? last_hidden : a.alloc_h(M * H);
        op_layernorm(c, final_ln, View(slab(L), H, H), View(lh, H, H), M, eps);
        if (pooled || text_embeds) {
            if (!eos_index) { set_error("clip: eos_index is required for pooled / text_embeds"); return 1; }
            half_t* p16 = pooled ? pooled : a.alloc_h((long)B * H);
            float* p32 = a.alloc_f((long)B * H);
            if (go && !c.err) c.err = launch_gather_rows(lh, H, eos_index, p16, p32, B, T, H, s);
            if (text_embeds) {
                if (!proj) { set_error("clip: text_embeds requested from a model without text_projection"); return 1; }
                float* e32 = a.alloc_f((long)B * cfg.projection_dim);
                if (go && !c.err) c.err = launch_small_linear(p32, H, proj, nullptr, e32, cfg.projection_dim, B, H,
                                                              cfg.projection_dim, 0, 0, s);
                if (go && !c.err) c.err = launch_f32_to_f16(e32, text_embeds, (long)B * cfg.projection_dim, s);
            }
        }
    }
    return c.err;
}

int CLIP::forward(const int* ids, const int* eos_index, half_t* hidden_states, half_t* last_hidden, half_t* pooled,
                  half_t* text_embeds, int B, int T, hipStream_t stream) {
    if (!finalized) { set_error("clip: forward before finalize"); return 2; }
    if (B <= 0 || T <= 0 || T > cfg.max_positions) { set_error("clip: bad shape (T must be <= max_positions)"); return 1; }
    const long key = ((long)B << 32) ^ ((long)T << 8) ^ (hidden_states ? 1 : 0) ^ (last_hidden ? 2 : 0) ^
                     (pooled ? 4 : 0) ^ (text_embeds ? 8 : 0);
    if (key != planned_key) {
        Ctx dry{&arena, stream, true};
        arena.begin(true);
        int rc = run(dry, ids, eos_index, hidden_states, last_hidden, pooled, text_embeds, B, T);
        if (rc) return rc;
        if (arena.peak() > arena.capacity()) {
            SD_HIP_CHECK(hipDeviceSynchronize());
            if ((rc = arena.reserve(arena.peak()))) return rc;
        }
        planned_key = key;
    }
    Ctx ctx{&arena, stream, false};
    arena.begin(false);
    int rc = run(ctx, ids, eos_index, hidden_states, last_hidden, pooled, text_embeds, B, T);
    if (!rc && arena.overflow()) { set_error("clip: workspace overflow (planner bug)"); return 2; }
    return rc;
}

int CLIP::final_layer_norm(const half_t* x, half_t* y, long rows, hipStream_t stream) {
    if (!finalized) { set_error("clip: final_layer_norm before finalize"); return 2; }
    return launch_layernorm(x, cfg.hidden_size, final_ln.gamma, final_ln.beta, y, cfg.hidden_size, rows, cfg.hidden_size,
                            cfg.layer_norm_eps, stream);
}

}  // namespace sd
