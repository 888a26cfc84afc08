// AutoencoderKL decoder / encoder graph on the gfx950 kernels.
//
// What it computes: diffusers 0.27.2 AutoencoderKL.decode (reference call site
// /root/reference/pipelines/sd_unified_pipeline.py:523) and .encode up to the moments (:1027-1032);
// weight names per /root/reference/scripts/convert_from_A1111.py:572-677.  All norms eps = 1e-6,
// mid attention = one head over all channels with biased q/k/v.  Runs NHWC fp16 with the nearest-2x
// upsample folded into the following conv's gather (no upsampled tensor is ever written).
#include "model.h"
#include <cmath>

namespace sd {

void declare_resnet(WeightStore& ws, const std::string& p, int cin, int cout, int temb);

namespace {
constexpr float kVaeEps = 1e-6f;

void declare_attn(WeightStore& ws, const std::string& p, int c) {
    ws.declare(p + ".group_norm.weight", {c});
    ws.declare(p + ".group_norm.bias", {c});
    for (const char* n : {"to_q", "to_k", "to_v", "to_out.0"}) {
        ws.declare(p + "." + n + ".weight", {c, c});
        ws.declare(p + "." + n + ".bias", {c});
    }
}
}  // namespace

VAE::VAE(const sd_vae_config& c) : cfg(c) {
    const int nb = cfg.num_blocks;
    const int* boc = cfg.block_out_channels;
    const int lc = cfg.latent_channels;
    // encoder (declared first: same order as stablediffusion_amd/weights.py::vae_manifest)
    ws.declare("encoder.conv_in.weight", {boc[0], cfg.in_channels, 3, 3});
    ws.declare("encoder.conv_in.bias", {boc[0]});
    int out_ch = boc[0];
    for (int i = 0; i < nb; ++i) {
        const int in_ch = out_ch;
        out_ch = boc[i];
        const std::string p = "encoder.down_blocks." + std::to_string(i);
        for (int j = 0; j < cfg.layers_per_block; ++j)
            declare_resnet(ws, p + ".resnets." + std::to_string(j), j == 0 ? in_ch : out_ch, out_ch, 0);
        if (i != nb - 1) {
            ws.declare(p + ".downsamplers.0.conv.weight", {out_ch, out_ch, 3, 3});
            ws.declare(p + ".downsamplers.0.conv.bias", {out_ch});
        }
    }
    const int top = boc[nb - 1];
    declare_resnet(ws, "encoder.mid_block.resnets.0", top, top, 0);
    declare_attn(ws, "encoder.mid_block.attentions.0", top);
    declare_resnet(ws, "encoder.mid_block.resnets.1", top, top, 0);
    ws.declare("encoder.conv_norm_out.weight", {top});
    ws.declare("encoder.conv_norm_out.bias", {top});
    ws.declare("encoder.conv_out.weight", {2 * lc, top, 3, 3});
    ws.declare("encoder.conv_out.bias", {2 * lc});
    ws.declare("quant_conv.weight", {2 * lc, 2 * lc, 1, 1});
    ws.declare("quant_conv.bias", {2 * lc});
    // decoder
    ws.declare("post_quant_conv.weight", {lc, lc, 1, 1});
    ws.declare("post_quant_conv.bias", {lc});
    ws.declare("decoder.conv_in.weight", {top, lc, 3, 3});
    ws.declare("decoder.conv_in.bias", {top});
    declare_resnet(ws, "decoder.mid_block.resnets.0", top, top, 0);
    declare_attn(ws, "decoder.mid_block.attentions.0", top);
    declare_resnet(ws, "decoder.mid_block.resnets.1", top, top, 0);
    out_ch = top;
    for (int i = 0; i < nb; ++i) {
        const int prev = out_ch;
        out_ch = boc[nb - 1 - i];
        const std::string p = "decoder.up_blocks." + std::to_string(i);
        for (int j = 0; j < cfg.layers_per_block + 1; ++j)
            declare_resnet(ws, p + ".resnets." + std::to_string(j), j == 0 ? prev : out_ch, out_ch, 0);
        if (i != nb - 1) {
            ws.declare(p + ".upsamplers.0.conv.weight", {out_ch, out_ch, 3, 3});
            ws.declare(p + ".upsamplers.0.conv.bias", {out_ch});
        }
    }
    ws.declare("decoder.conv_norm_out.weight", {boc[0]});
    ws.declare("decoder.conv_norm_out.bias", {boc[0]});
    ws.declare("decoder.conv_out.weight", {cfg.out_channels, boc[0], 3, 3});
    ws.declare("decoder.conv_out.bias", {cfg.out_channels});
}

int VAE::pack_resnet(const std::string& p, Resnet* r) {
    int rc;
    if ((rc = ws.pack_norm(p + ".norm1", &r->n1))) return rc;
    if ((rc = ws.pack_conv(p + ".conv1", &r->c1))) return rc;
    if ((rc = ws.pack_norm(p + ".norm2", &r->n2))) return rc;
    if ((rc = ws.pack_conv(p + ".conv2", &r->c2))) return rc;
    r->cin = r->c1.cin; r->cout = r->c1.cout;
    r->has_sc = ws.raw(p + ".conv_shortcut.weight") != nullptr;
    if (r->has_sc && (rc = ws.pack_conv(p + ".conv_shortcut", &r->sc))) return rc;
    return 0;
}

int VAE::pack_attn(const std::string& p, VaeAttn* a) {
    int rc;
    if ((rc = ws.pack_norm(p + ".group_norm", &a->gn))) return rc;
    if ((rc = ws.pack_rows({p + ".to_q.weight", p + ".to_k.weight", p + ".to_v.weight"},
                           {p + ".to_q.bias", p + ".to_k.bias", p + ".to_v.bias"}, &a->qkv))) return rc;
    if ((rc = ws.pack_conv(p + ".to_out.0", &a->out))) return rc;
    a->C = a->out.cout;
    return 0;
}

int VAE::pack_pointwise(const std::string& p, half_t** w, float** b) {
    const RawTensor* rw = ws.raw(p + ".weight");
    const RawTensor* rb = ws.raw(p + ".bias");
    if (!rw || !rb) { set_error("missing weight: " + p); return 2; }
    *w = static_cast<half_t*>(ws.dmalloc((size_t)rw->numel * sizeof(half_t)));
    *b = static_cast<float*>(ws.dmalloc((size_t)rb->numel * sizeof(float)));
    if (!*w || !*b) { set_error("hipMalloc failed"); return 3; }
    SD_HIP_CHECK(hipMemcpy(*w, rw->dev, (size_t)rw->numel * sizeof(half_t), hipMemcpyDeviceToDevice));
    int rc = launch_f16_to_f32(rb->dev, *b, rb->numel, 0);
    return rc;
}

int VAE::finalize() {
    if (finalized) return 0;
    std::string missing;
    if (!ws.complete(&missing)) { set_error("finalize: weight not set: " + missing); return 2; }
    const int nb = cfg.num_blocks;
    int rc;
    // decoder
    if ((rc = pack_pointwise("post_quant_conv", &pq_w, &pq_b))) return rc;
    if ((rc = ws.pack_conv("decoder.conv_in", &d_conv_in))) return rc;
    if ((rc = pack_resnet("decoder.mid_block.resnets.0", &d_mid0))) return rc;
    if ((rc = pack_attn("decoder.mid_block.attentions.0", &d_attn))) return rc;
    if ((rc = pack_resnet("decoder.mid_block.resnets.1", &d_mid1))) return rc;
    d_up.assign((size_t)nb, {}); d_us.assign((size_t)nb, ConvW());
    for (int i = 0; i < nb; ++i) {
        const std::string p = "decoder.up_blocks." + std::to_string(i);
        d_up[i].resize((size_t)cfg.layers_per_block + 1);
        for (int j = 0; j < cfg.layers_per_block + 1; ++j)
            if ((rc = pack_resnet(p + ".resnets." + std::to_string(j), &d_up[i][j]))) return rc;
        if (i != nb - 1 && (rc = ws.pack_conv(p + ".upsamplers.0.conv", &d_us[i]))) return rc;
    }
    if ((rc = ws.pack_norm("decoder.conv_norm_out", &d_norm_out))) return rc;
    if ((rc = ws.pack_conv("decoder.conv_out", &d_conv_out))) return rc;
    // encoder
    if ((rc = pack_pointwise("quant_conv", &q_w, &q_b))) return rc;
    if ((rc = ws.pack_conv("encoder.conv_in", &e_conv_in))) return rc;
    e_down.assign((size_t)nb, {}); e_ds.assign((size_t)nb, ConvW());
    for (int i = 0; i < nb; ++i) {
        const std::string p = "encoder.down_blocks." + std::to_string(i);
        e_down[i].resize((size_t)cfg.layers_per_block);
        for (int j = 0; j < cfg.layers_per_block; ++j)
            if ((rc = pack_resnet(p + ".resnets." + std::to_string(j), &e_down[i][j]))) return rc;
        if (i != nb - 1 && (rc = ws.pack_conv(p + ".downsamplers.0.conv", &e_ds[i]))) return rc;
    }
    if ((rc = pack_resnet("encoder.mid_block.resnets.0", &e_mid0))) return rc;
    if ((rc = pack_attn("encoder.mid_block.attentions.0", &e_attn))) return rc;
    if ((rc = pack_resnet("encoder.mid_block.resnets.1", &e_mid1))) return rc;
    if ((rc = ws.pack_norm("encoder.conv_norm_out", &e_norm_out))) return rc;
    if ((rc = ws.pack_conv("encoder.conv_out", &e_conv_out))) return rc;
    SD_HIP_CHECK(hipDeviceSynchronize());
    ws.free_raw();
    finalized = true;
    return 0;
}

void VAE::run_attn(Ctx& c, const VaeAttn& at, View x, int N, int H, int W, View out, const GnStatBuf* x_stats,
                   GnStatBuf** out_stats, float stream_scale) {
    Arena& a = *c.arena;
    const size_t mk = a.mark();
    const int C = at.C;
    const long M = (long)N * H * W;
    View hn(a.alloc_h(M * C), C, C);
    op_groupnorm(c, at.gn, x, hn, N, (long)H * W, cfg.norm_num_groups, kVaeEps * stream_scale * stream_scale, 0, x_stats);
    View qkv(a.alloc_h(M * 3 * C), 3 * C, 3 * C);
    op_conv(c, at.qkv, hn, N, H, W, qkv);
    View o(a.alloc_h(M * C), C, C);
    op_attention(c, qkv.slice(0, C), qkv.slice(C, C), qkv.slice(2 * C, C), o, N, H * W, H * W, 1, C);
    ConvFuse fo;
    fo.gn_out = (out_stats && gn_wants_stats((long)H * W, C, cfg.norm_num_groups)) ? ctx_gnbuf(c) : nullptr;
    fo.gn_groups = cfg.norm_num_groups;
    fo.acc_scale = stream_scale; fo.bias_scale = stream_scale;      // the projection's output joins the (scaled) residual stream
    op_conv(c, at.out, o, N, H, W, out, 1, 0, nullptr, 0, &x, 0, -1, 0, &fo);
    if (out_stats) *out_stats = fo.gn_out;
    a.release(mk);
}

int VAE::run_decode(Ctx& c, const half_t* z, half_t* img, int B, int h, int w) {
    Arena& a = *c.arena;
    const int nb = cfg.num_blocks;
    const int* boc = cfg.block_out_channels;
    const int G = cfg.norm_num_groups;
    const int lc = cfg.latent_channels;
    hipStream_t s = c.stream;
    const bool go = !c.dry;
    const int top = boc[nb - 1];
    long M = (long)B * h * w;

    // GroupNorm summaries ride from each convolution's epilogue to the GroupNorm that follows (engine.h)
    ctx_gnpool_init(c, B, (long)h * w << (2 * (nb - 1)), G);
    GnStatBuf* xs = nullptr;
    auto gn_fuse = [&](ConvFuse& f, long HW, int C) {
        f.gn_out = gn_wants_stats(HW, C, G) ? ctx_gnbuf(c) : nullptr;
        f.gn_groups = G;
    };

    // post_quant_conv (pointwise on NCHW, 4 channels) + conv_in via im2col (K = 36 -> 64)
    View x(a.alloc_h(M * top), top, top);
    {
        const size_t mk = a.mark();
        half_t* pq = a.alloc_h(M * lc);
        half_t* col = a.alloc_h(M * d_conv_in.K);
        if (go && !c.err) c.err = launch_pointwise_nchw(z, pq_w, pq_b, pq, B, lc, lc, (long)h * w, s);
        if (go && !c.err) c.err = launch_im2col_nchw3x3(pq, col, B, lc, h, w, (int)d_conv_in.K, s);
        ConvW pw = d_conv_in; pw.ks = 1;
        ConvFuse f;
        gn_fuse(f, (long)h * w, top);
        op_conv(c, pw, View(col, d_conv_in.K, (int)d_conv_in.K), B, h, w, x, 1, 0, nullptr, 0, nullptr, 0, -1, 0, &f);
        xs = f.gn_out;
        a.release(mk);
    }
    // ping-pong activation buffers sized for the largest level, allocated once
    // (every resnet's temporaries are stack-released inside run_resnet)
    View y(a.alloc_h(M * top), top, top);
    run_resnet(c, d_mid0, x, B, h, w, y, G, kVaeEps, nullptr, 0, xs, &xs);
    run_attn(c, d_attn, y, B, h, w, x, xs, &xs);
    run_resnet(c, d_mid1, x, B, h, w, y, G, kVaeEps, nullptr, 0, xs, &xs);
    View cur = y;
    for (int i = 0; i < nb; ++i) {
        for (int j = 0; j < cfg.layers_per_block + 1; ++j) {
            const Resnet& r = d_up[i][j];
            View nxt(a.alloc_h(M * r.cout), r.cout, r.cout);
            run_resnet(c, r, cur, B, h, w, nxt, G, kVaeEps, nullptr, 0, xs, &xs);
            cur = nxt;
        }
        if (i != nb - 1) {
            const int C = d_us[i].cout;
            View nxt(a.alloc_h(M * 4 * C), C, C);
            ConvFuse f;
            gn_fuse(f, (long)h * w * 4, C);
            op_conv(c, d_us[i], cur, B, h, w, nxt, 1, 1, nullptr, 0, nullptr, 0, -1, 0, &f);
            xs = f.gn_out;
            h *= 2; w *= 2; M *= 4;
            cur = nxt;
        }
    }
    {
        const int C = boc[0];
        View hn(a.alloc_h(M * C), C, C);
        op_groupnorm(c, d_norm_out, cur, hn, B, (long)h * w, G, kVaeEps, 1, xs);
        if (cfg.out_channels <= 4 && d_conv_out.ks == 3 && d_conv_out.K == 9L * C && M >= kSmallCoutMinPixels) {
            if (go && !c.err) {
                prof_open(s, "conv3x3_small_cout_kernel", 2.0 * M * cfg.out_channels * 9.0 * C, 2.0 * M * (C + cfg.out_channels));
                c.err = launch_conv3x3_small_cout(hn.p, hn.ld, d_conv_out.w, d_conv_out.K, d_conv_out.bias, img, B, h, w, C,
                                                  cfg.out_channels, s);
                prof_close(s);
            }
        } else {
            View o(a.alloc_h(M * cfg.out_channels), cfg.out_channels, cfg.out_channels);
            op_conv(c, d_conv_out, hn, B, h, w, o);
            if (go && !c.err) c.err = launch_nhwc_to_nchw(o.p, o.ld, img, B, (long)h * w, cfg.out_channels, s);
        }
    }
    return c.err;
}

// stream_scale s = 2^-encode_shift: every activation between conv_in and conv_norm_out is stored s times its true value
// (conv_in and the convolutions that write the residual stream emit s * (conv + bias); those that read it -- shortcuts,
// downsamplers -- scale only their bias; every GroupNorm runs with eps * s^2).  The function computed is the same.
int VAE::run_encode(Ctx& c, const half_t* img, half_t* moments, int B, int H, int W, float stream_scale) {
    const float ss = stream_scale;
    Arena& a = *c.arena;
    const int nb = cfg.num_blocks;
    const int* boc = cfg.block_out_channels;
    const int G = cfg.norm_num_groups;
    const int lc2 = 2 * cfg.latent_channels;
    hipStream_t s = c.stream;
    const bool go = !c.dry;
    int h = H, w = W;
    long M = (long)B * h * w;
    ctx_gnpool_init(c, B, (long)H * W, G);
    GnStatBuf* xs = nullptr;
    auto gn_fuse = [&](ConvFuse& f, long HW, int C) {
        f.gn_out = gn_wants_stats(HW, C, G) ? ctx_gnbuf(c) : nullptr;
        f.gn_groups = G;
    };
    View cur(a.alloc_h(M * boc[0]), boc[0], boc[0]);
    {
        const size_t mk = a.mark();
        half_t* col = a.alloc_h(M * e_conv_in.K);
        if (go && !c.err) c.err = launch_im2col_nchw3x3(img, col, B, cfg.in_channels, h, w, (int)e_conv_in.K, s);
        ConvW pw = e_conv_in; pw.ks = 1;
        ConvFuse f;
        gn_fuse(f, (long)h * w, boc[0]);
        f.acc_scale = ss; f.bias_scale = ss;
        op_conv(c, pw, View(col, e_conv_in.K, (int)e_conv_in.K), B, h, w, cur, 1, 0, nullptr, 0, nullptr, 0, -1, 0, &f);
        xs = f.gn_out;
        a.release(mk);
    }
    for (int i = 0; i < nb; ++i) {
        for (int j = 0; j < cfg.layers_per_block; ++j) {
            const Resnet& r = e_down[i][j];
            View nxt(a.alloc_h(M * r.cout), r.cout, r.cout);
            run_resnet(c, r, cur, B, h, w, nxt, G, kVaeEps, nullptr, 0, xs, &xs, ss);
            cur = nxt;
        }
        if (i != nb - 1) {
            const int C = e_ds[i].cout;
            View nxt(a.alloc_h(M / 4 * C), C, C);
            ConvFuse f;
            gn_fuse(f, (long)(h / 2) * (w / 2), C);
            f.bias_scale = ss;                   // (reads the scaled stream)
            op_conv(c, e_ds[i], cur, B, h, w, nxt, 2, 0, nullptr, 0, nullptr, 0, /*pad=*/0, 0, &f);
            xs = f.gn_out;
            h /= 2; w /= 2; M /= 4;
            cur = nxt;
        }
    }
    const int top = boc[nb - 1];
    View t0(a.alloc_h(M * top), top, top), t1(a.alloc_h(M * top), top, top);
    run_resnet(c, e_mid0, cur, B, h, w, t0, G, kVaeEps, nullptr, 0, xs, &xs, ss);
    run_attn(c, e_attn, t0, B, h, w, t1, xs, &xs, ss);
    run_resnet(c, e_mid1, t1, B, h, w, t0, G, kVaeEps, nullptr, 0, xs, &xs, ss);
    op_groupnorm(c, e_norm_out, t0, t1, B, (long)h * w, G, kVaeEps * ss * ss, 1, xs);
    View o(a.alloc_h(M * lc2), lc2, lc2);
    op_conv(c, e_conv_out, t1, B, h, w, o);
    half_t* o_nchw = a.alloc_h(M * lc2);
    if (go && !c.err) c.err = launch_nhwc_to_nchw(o.p, o.ld, o_nchw, B, (long)h * w, lc2, s);
    if (go && !c.err) c.err = launch_pointwise_nchw(o_nchw, q_w, q_b, moments, B, lc2, lc2, (long)h * w, s);
    return c.err;
}

int VAE::decode(const half_t* z, half_t* img, int B, int h, int w, hipStream_t stream) {
    if (!finalized) { set_error("vae: decode before finalize"); return 2; }
    if (B <= 0 || h <= 0 || w <= 0) { set_error("vae: bad shape"); return 1; }
    const long key = ((long)B << 40) ^ ((long)h << 20) ^ (long)w;
    if (key != planned_key) {
        Ctx dry{&arena, stream, true};
        arena.begin(true);
        int rc = run_decode(dry, z, img, B, h, w);
        if (rc) return rc;
        if (arena.peak() > arena.capacity()) {
            SD_HIP_CHECK(hipDeviceSynchronize());
            if ((rc = arena.reserve(arena.peak()))) return rc;
        }
        planned_key = key;
    }
    Ctx ctx{&arena, stream, false};
    arena.begin(false);
    int rc = run_decode(ctx, z, img, B, h, w);
    if (!rc && arena.overflow()) { set_error("vae: workspace overflow (planner bug)"); return 2; }
    return rc;
}

int VAE::encode(const half_t* img, half_t* moments, int B, int H, int W, hipStream_t stream) {
    if (!finalized) { set_error("vae: encode before finalize"); return 2; }
    const int div = 1 << (cfg.num_blocks - 1);
    if (B <= 0 || H % div != 0 || W % div != 0) { set_error("vae: H and W must be divisible by 2^(blocks-1)"); return 1; }
    const long key = (((long)B << 40) ^ ((long)H << 20) ^ (long)W) | (1L << 62);
    if (key != planned_key) {
        Ctx dry{&arena, stream, true};
        arena.begin(true);
        int rc = run_encode(dry, img, moments, B, H, W, ldexpf(1.f, -encode_shift));
        if (rc) return rc;
        if (arena.peak() > arena.capacity()) {
            SD_HIP_CHECK(hipDeviceSynchronize());
            if ((rc = arena.reserve(arena.peak()))) return rc;
        }
        planned_key = key;
    }
    Ctx ctx{&arena, stream, false};
    arena.begin(false);
    int rc = run_encode(ctx, img, moments, B, H, W, ldexpf(1.f, -encode_shift));
    if (!rc && arena.overflow()) { set_error("vae: workspace overflow (planner bug)"); return 2; }
    return rc;
}

}  // namespace sd
