// Arena, weight store and op wrappers of the denoise engine (host side, HIP runtime only).
#include "engine.h"

#include <atomic>
#include <cstdio>
#include <mutex>
#include <cstdlib>
#include <cstring>

namespace sd {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }

// ------------------------------------------------------------------------------------------ Arena
Arena::~Arena() {
    if (base_) (void)hipFree(base_);
}

int Arena::reserve(size_t bytes) {
    if (bytes <= cap_) return 0;
    if (base_) { (void)hipFree(base_); base_ = nullptr; cap_ = 0; }
    SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&base_), bytes));
    cap_ = bytes;
    return 0;
}

void* Arena::alloc(size_t bytes) {
    const size_t aligned = (bytes + 255) & ~size_t(255);
    const size_t at = off_;
    off_ += aligned;
    if (off_ > peak_) peak_ = off_;
    if (dry_) return reinterpret_cast<void*>(uintptr_t(0x1000) + at);  // never dereferenced
    if (off_ > cap_) { overflow_ = true; return base_; }
    return base_ + at;
}

// ------------------------------------------------------------------------------------ WeightStore
WeightStore::~WeightStore() {
    free_raw();
    for (void* p : owned_) (void)hipFree(p);
}

void* WeightStore::dmalloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return nullptr;
    owned_.push_back(p);
    packed_bytes_ += (int64_t)bytes;
    return p;
}

void WeightStore::declare(const std::string& key, std::vector<int64_t> shape) {
    RawTensor t;
    t.shape = std::move(shape);
    t.numel = 1;
    for (auto s : t.shape) t.numel *= s;
    tensors.emplace(key, std::move(t));
    order.push_back(key);
}

int WeightStore::set(const std::string& key, const void* data, const int64_t* shape, int ndim, int dtype) {
    auto it = tensors.find(key);
    if (it == tensors.end()) { set_error("unknown weight key: " + key); return 1; }
    RawTensor& t = it->second;
    if ((int)t.shape.size() != ndim) { set_error("rank mismatch for " + key); return 1; }
    for (int i = 0; i < ndim; ++i)
        if (t.shape[i] != shape[i]) {
            set_error("shape mismatch for " + key + " at dim " + std::to_string(i) + ": expected " +
                      std::to_string(t.shape[i]) + ", got " + std::to_string(shape[i]));
            return 1;
        }
    if (!t.dev) SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&t.dev), (size_t)t.numel * sizeof(half_t)));
    if (dtype == 0) {
        SD_HIP_CHECK(hipMemcpy(t.dev, data, (size_t)t.numel * sizeof(half_t), hipMemcpyDefault));
    } else if (dtype == 1) {
        float* tmp = nullptr;
        SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)t.numel * sizeof(float)));
        hipError_t e = hipMemcpy(tmp, data, (size_t)t.numel * sizeof(float), hipMemcpyDefault);
        int rc = 0;
        if (e == hipSuccess) rc = launch_f32_to_f16(tmp, t.dev, t.numel, 0);
        if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(0);
        (void)hipFree(tmp);
        if (e != hipSuccess) { set_error(std::string("set_weight copy: ") + hipGetErrorString(e)); return 3; }
        if (rc) return rc;
    } else {
        set_error("unsupported dtype code");
        return 1;
    }
    return 0;
}

bool WeightStore::complete(std::string* missing) const {
    for (const auto& k : order) {
        auto it = tensors.find(k);
        if (it == tensors.end() || !it->second.dev) {
            if (missing) *missing = k;
            return false;
        }
    }
    return true;
}

const RawTensor* WeightStore::raw(const std::string& key) const {
    auto it = tensors.find(key);
    if (it == tensors.end() || !it->second.dev) return nullptr;
    return &it->second;
}

void WeightStore::free_raw() {
    for (auto& kv : tensors)
        if (kv.second.dev) { (void)hipFree(kv.second.dev); kv.second.dev = nullptr; }
}

int WeightStore::host_floats(const std::string& key, std::vector<float>* out) const {
    const RawTensor* t = raw(key);
    if (!t) { set_error("missing weight: " + key); return 2; }
    std::vector<half_t> h((size_t)t->numel);
    SD_HIP_CHECK(hipMemcpy(h.data(), t->dev, (size_t)t->numel * sizeof(half_t), hipMemcpyDeviceToHost));
    out->resize((size_t)t->numel);
    for (long i = 0; i < t->numel; ++i) (*out)[(size_t)i] = (float)h[(size_t)i];
    return 0;
}

static long round_up(long a, long b) { return (a + b - 1) / b * b; }

static int upload_bias(WeightStore* ws, const std::vector<float>& host, int cout, float** dst) {
    const long padded = round_up(cout, kWeightRowPad);
    std::vector<float> buf((size_t)padded, 0.f);
    std::memcpy(buf.data(), host.data(), (size_t)cout * sizeof(float));
    *dst = static_cast<float*>(ws->dmalloc((size_t)padded * sizeof(float)));
    if (!*dst) { set_error("hipMalloc failed (bias)"); return 3; }
    SD_HIP_CHECK(hipMemcpy(*dst, buf.data(), (size_t)padded * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int WeightStore::pack_conv(const std::string& prefix, ConvW* out, bool has_bias) {
    const RawTensor* w = raw(prefix + ".weight");
    if (!w) { set_error("missing weight: " + prefix + ".weight"); return 2; }
    const int O = (int)w->shape[0], I = (int)w->shape[1];
    const int KH = w->shape.size() == 4 ? (int)w->shape[2] : 1;
    const int KW = w->shape.size() == 4 ? (int)w->shape[3] : 1;
    const long K = round_up((long)KH * KW * I, 64);
    const long rows = round_up(O, kWeightRowPad);
    out->cin = I; out->cout = O; out->ks = KH; out->K = K;
    out->w = static_cast<half_t*>(dmalloc((size_t)rows * K * sizeof(half_t)));
    if (!out->w) { set_error("hipMalloc failed (weights)"); return 3; }
    SD_HIP_CHECK(hipMemsetAsync(out->w, 0, (size_t)rows * K * sizeof(half_t), 0));
    int rc = launch_pack_conv(w->dev, out->w, O, I, KH, KW, K, 0);
    if (rc) return rc;
    out->bias = nullptr;
    if (has_bias) {
        std::vector<float> b;
        rc = host_floats(prefix + ".bias", &b);
        if (rc) return rc;
        rc = upload_bias(this, b, O, &out->bias);
        if (rc) return rc;
    }
    return 0;
}

int WeightStore::pack_rows(const std::vector<std::string>& wkeys, const std::vector<std::string>& bkeys,
                           ConvW* out) {
    long total = 0;
    int I = -1;
    for (const auto& k : wkeys) {
        const RawTensor* w = raw(k);
        if (!w) { set_error("missing weight: " + k); return 2; }
        if (I < 0) I = (int)w->shape[1];
        if ((int)w->shape[1] != I) { set_error("pack_rows: inner dims differ at " + k); return 1; }
        total += w->shape[0];
    }
    if (I % 64 != 0) { set_error("pack_rows: inner dim must be a multiple of 64"); return 1; }
    const long rows = round_up(total, kWeightRowPad);
    out->cin = I; out->cout = (int)total; out->ks = 1; out->K = I;
    out->w = static_cast<half_t*>(dmalloc((size_t)rows * I * sizeof(half_t)));
    if (!out->w) { set_error("hipMalloc failed (weights)"); return 3; }
    SD_HIP_CHECK(hipMemsetAsync(out->w, 0, (size_t)rows * I * sizeof(half_t), 0));
    long r = 0;
    for (const auto& k : wkeys) {
        const RawTensor* w = raw(k);
        SD_HIP_CHECK(hipMemcpyAsync(out->w + r * I, w->dev, (size_t)w->numel * sizeof(half_t),
                                    hipMemcpyDeviceToDevice, 0));
        r += w->shape[0];
    }
    out->bias = nullptr;
    if (!bkeys.empty()) {
        std::vector<float> all;
        for (const auto& k : bkeys) {
            std::vector<float> b;
            int rc = host_floats(k, &b);
            if (rc) return rc;
            all.insert(all.end(), b.begin(), b.end());
        }
        if ((long)all.size() != total) { set_error("pack_rows: bias length mismatch"); return 1; }
        int rc = upload_bias(this, all, (int)total, &out->bias);
        if (rc) return rc;
    }
    return 0;
}

// GEGLU projection [8C][C]: rows 0..4C-1 = hidden, 4C..8C-1 = gate (diffusers GEGLU: hidden, gate =
// proj.chunk(2)).  Packed so that every 128-row group holds 64 hidden rows followed by their 64
// gate rows -- the igemm epilogue then finds both halves of out = hidden * gelu(gate) in one tile.
int WeightStore::pack_geglu(const std::string& prefix, ConvW* out) {
    const RawTensor* w = raw(prefix + ".weight");
    if (!w) { set_error("missing weight: " + prefix + ".weight"); return 2; }
    const long O = w->shape[0], I = w->shape[1];
    const long half_rows = O / 2;
    if (half_rows % 64 != 0 || I % 64 != 0) { set_error("pack_geglu: dims must be multiples of 64"); return 1; }
    const long rows = round_up(O, kWeightRowPad);
    out->cin = (int)I; out->cout = (int)O; out->ks = 1; out->K = I;
    out->w = static_cast<half_t*>(dmalloc((size_t)rows * I * sizeof(half_t)));
    if (!out->w) { set_error("hipMalloc failed (weights)"); return 3; }
    SD_HIP_CHECK(hipMemsetAsync(out->w, 0, (size_t)rows * I * sizeof(half_t), 0));
    std::vector<float> b, bp((size_t)O);
    int rc = host_floats(prefix + ".bias", &b);
    if (rc) return rc;
    for (long blk = 0; blk < half_rows / 64; ++blk) {
        SD_HIP_CHECK(hipMemcpyAsync(out->w + (blk * 128) * I, w->dev + (blk * 64) * I,
                                    (size_t)64 * I * sizeof(half_t), hipMemcpyDeviceToDevice, 0));
        SD_HIP_CHECK(hipMemcpyAsync(out->w + (blk * 128 + 64) * I, w->dev + (half_rows + blk * 64) * I,
                                    (size_t)64 * I * sizeof(half_t), hipMemcpyDeviceToDevice, 0));
        for (int e = 0; e < 64; ++e) {
            bp[(size_t)(blk * 128 + e)] = b[(size_t)(blk * 64 + e)];
            bp[(size_t)(blk * 128 + 64 + e)] = b[(size_t)(half_rows + blk * 64 + e)];
        }
    }
    return upload_bias(this, bp, (int)O, &out->bias);
}

int WeightStore::pack_norm(const std::string& prefix, NormW* out) {
    std::vector<float> g, b;
    int rc = host_floats(prefix + ".weight", &g);
    if (rc) return rc;
    rc = host_floats(prefix + ".bias", &b);
    if (rc) return rc;
    out->C = (int)g.size();
    out->gamma = static_cast<float*>(dmalloc(g.size() * sizeof(float)));
    out->beta = static_cast<float*>(dmalloc(b.size() * sizeof(float)));
    if (!out->gamma || !out->beta) { set_error("hipMalloc failed (norm)"); return 3; }
    SD_HIP_CHECK(hipMemcpy(out->gamma, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice));
    SD_HIP_CHECK(hipMemcpy(out->beta, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    out->gb = nullptr;
    if (g.size() % 64 == 0) {          // packed per 64 channels for the convolution-fused GroupNorm
        std::vector<float> gb(2 * g.size());
        for (size_t blk = 0; blk < g.size() / 64; ++blk)
            for (int e = 0; e < 64; ++e) {
                gb[blk * 128 + e] = g[blk * 64 + e];
                gb[blk * 128 + 64 + e] = b[blk * 64 + e];
            }
        out->gb = static_cast<float*>(dmalloc(gb.size() * sizeof(float)));
        if (!out->gb) { set_error("hipMalloc failed (norm)"); return 3; }
        SD_HIP_CHECK(hipMemcpy(out->gb, gb.data(), gb.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

int WeightStore::fold_ln(ConvW* w, const NormW& ln, int rows_scaled, float row_scale) {
    if (ln.C != w->K || w->ks != 1) { set_error("fold_ln: LayerNorm width must equal the linear's K"); return 1; }
    const long padded = round_up(w->cout, kWeightRowPad);
    float* nb = static_cast<float*>(dmalloc((size_t)padded * sizeof(float)));
    w->wsum = static_cast<float*>(dmalloc((size_t)padded * sizeof(float)));
    if (!nb || !w->wsum) { set_error("hipMalloc failed (fold_ln)"); return 3; }
    SD_HIP_CHECK(hipMemsetAsync(nb, 0, (size_t)padded * sizeof(float), 0));
    SD_HIP_CHECK(hipMemsetAsync(w->wsum, 0, (size_t)padded * sizeof(float), 0));
    int rc = launch_ln_fold(w->w, w->K, w->cout, ln.gamma, ln.beta, w->bias, nb, w->wsum, rows_scaled, row_scale, 0);
    if (rc) return rc;
    w->bias = nb;
    return 0;
}

// ---------------------------------------------------------------------------------------- profiler
namespace {
// The profiler is process-wide (bench.py brackets one forward at a time); the record list is guarded so that
// two handles driven from two threads cannot corrupt it -- their rows would interleave, nothing worse.
struct ProfRec { std::string name; double flops, bytes; hipEvent_t a, b; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
}  // namespace
void prof_enable(bool on) { g_prof_on = on; }
bool prof_enabled() { return g_prof_on; }
void prof_open(hipStream_t s, const char* kernel, double flops, double bytes) {
    if (!g_prof_on) return;
    ProfRec r{kernel, flops, bytes, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
}
void prof_close(hipStream_t s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.empty()) return;
    (void)hipEventRecord(g_prof.back().b, s);
}
int prof_collect(std::map<std::string, ProfAgg>* out) {
    SD_HIP_CHECK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ProfAgg& a = (*out)[r.name];
            a.flops += r.flops; a.bytes += r.bytes; a.ms += ms; a.launches += 1;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return 0;
}

void ctx_gnpool_init(Ctx& c, int N, long HW_max, int G) {
    static const bool off = getenv("SD_NO_GN_EPILOGUE") != nullptr;    // A/B switch: GroupNorm runs its own statistics pass
    c.gn_groups = G;
    c.gn_next = 0;
    for (int i = 0; i < Ctx::kGnPool; ++i) {
        c.gnpool[i].buf = off ? nullptr : c.arena->alloc_f(gnstat_floats(N, HW_max, G));
        c.gnpool[i].st = GnStats();
    }
}

GnStatBuf* ctx_gnbuf(Ctx& c) {
    if (!c.gnpool[0].buf) return nullptr;
    GnStatBuf* b = &c.gnpool[c.gn_next];
    c.gn_next = (c.gn_next + 1) % Ctx::kGnPool;
    b->st = GnStats();
    return b;
}

// -------------------------------------------------------------------------------------------- ops
void op_conv(Ctx& c, const ConvW& w, View x, int N, int H, int W, View y, int stride, int up,
             const float* rowadd, int rowadd_ld, const View* res, int geglu, int pad, int act, const ConvFuse* fuse) {
    const RowStat* ln_in = fuse ? fuse->ln_in : nullptr;
    const float ln_eps = fuse ? fuse->ln_eps : 0.f;
    RowStat* stat_out = fuse ? fuse->stat_out : nullptr;
    GnStatBuf* gn_out = fuse ? fuse->gn_out : nullptr;
    IGemmParams p;
    p.x = x.p; p.ldx = x.ld;
    p.w = w.w; p.bias = w.bias;
    p.rowadd = rowadd; p.rowadd_ld = rowadd_ld;
    p.res = res ? res->p : nullptr; p.ldres = res ? res->ld : 0;
    p.y = y.p; p.ldy = y.ld;
    p.N = N; p.H = H; p.W = W; p.Cin = (w.ks == 1) ? (int)w.K : w.cin;
    p.KS = w.ks; p.stride = stride; p.up = up;
    p.pad = pad >= 0 ? pad : (w.ks == 3 ? 1 : 0);
    const int IH = H << up, IW = W << up;
    if (stride == 1) {
        p.OH = IH; p.OW = IW;
    } else if (p.pad == 0) {
        // VAE encoder Downsample2D: F.pad (0,1,0,1) then stride-2 conv with padding 0 -> the
        // right / bottom zero column is the gather's ordinary bounds check.
        p.OH = (IH + 1 - w.ks) / stride + 1; p.OW = (IW + 1 - w.ks) / stride + 1;
    } else {
        p.OH = (IH + 2 * p.pad - w.ks) / stride + 1; p.OW = (IW + 2 * p.pad - w.ks) / stride + 1;
    }
    p.Cout = w.cout;
    p.M = N * p.OH * p.OW;
    p.K = (int)w.K;
    p.geglu = geglu;
    p.act = act;
    if (ln_in) {
        if (!w.wsum) { set_error("op_conv: LayerNorm statistics passed to a linear without folded weights"); c.err = 1; return; }
        p.ln_stat = ln_in->p; p.ln_parts = ln_in->parts; p.ln_C = (int)w.K; p.ln_eps = ln_eps; p.ln_wsum = w.wsum;
    }
    if (fuse && fuse->gn_in) {
        p.gni_part = fuse->gn_in_stats.part; p.gni_S = fuse->gn_in_stats.S; p.gni_rows = fuse->gn_in_stats.rows;
        p.gni_groups = fuse->gn_in_groups; p.gni_eps = fuse->gn_in_eps; p.gni_silu = fuse->gn_in_silu;
        p.gni_gb = fuse->gn_in->gb;
        if (c.dry) p.gni_part = reinterpret_cast<const float*>(8);      // planning pass: only "is set" matters (tile choice)
    }
    if (fuse && (fuse->acc_scale != 1.f || fuse->bias_scale != 1.f)) {
        p.acc_scale = fuse->acc_scale; p.bias_scale = fuse->bias_scale;
        if (!igemm2_scales_ok(p)) { set_error("op_conv: output scaling needs the LDS-DMA kernels (not GEGLU / weight-stationary)"); c.err = 1; return; }
    }
    const bool v2 = igemm2_supported(p);
    if (ln_in && !v2) { set_error("op_conv: the LayerNorm fold needs the LDS-DMA kernel"); c.err = 1; return; }
    if (gn_out) {
        int rows = 0;
        gn_out->st = GnStats();
        if (gn_out->buf && igemm2_emits_gnstats(p, fuse->gn_groups, &rows)) {
            p.gnstat_out = gn_out->buf;
            p.gn_groups = fuse->gn_groups;
            gn_out->st.part = gn_out->buf;
            gn_out->st.rows = rows;
            gn_out->st.S = p.OH * p.OW / rows;
        }
    }
    // (after the GroupNorm request: the tile choice, and with it the partial count, depends on it)
    bool own_stats = false;      // the GEMM's epilogue writes the row statistics itself
    if (stat_out) {
        int parts = 1;
        own_stats = igemm2_emits_rowstats(p, &parts);
        stat_out->parts = own_stats ? parts : 1;
        if (own_stats) p.rowstat_out = stat_out->p;
    }
    if (act && !v2) { set_error("op_conv: activation epilogue needs the LDS-DMA kernel (Cin % 64, Cout % 8)"); c.err = 1; return; }
    float* partial = nullptr;
    if (v2) {
        const long pf = igemm2_partial_floats(p);
        if (pf > 0) partial = c.arena->alloc_f(pf);
    }
    if (c.dry || c.err) return;
    if (prof_enabled()) {
        const double kreal = (double)w.ks * w.ks * w.cin;
        const double in_px = (double)N * H * W;
        const char* name = igemm_variant(p);
        static thread_local char nbuf[64], fbuf[56];
        if (v2) {
            int var, sp;
            igemm2_pick(p, &var, &sp);
            const bool pw = p.KS == 1 && p.stride == 1 && p.up == 0;
            // split-K launches keep the kernel's name: their bracket also covers the small
            // splitk_epilogue_kernel, so the reported rate is slightly pessimistic for them
            snprintf(nbuf, sizeof(nbuf), igemm2_name(var), pw ? "true" : "false");
            const bool pg = p.geglu && var != 13 && var != 14 && pgemm_geglu_supported(p);
            if (pg) snprintf(nbuf, sizeof(nbuf), "geglu_persist_kernel");
            (void)fbuf;
            // SD_PROF_SHAPES=1 (tools/profile_layers.py): one row per (tile variant, split-K, problem shape,
            // fused extras) instead of one per kernel instantiation
            static const bool by_shape = getenv("SD_PROF_SHAPES") != nullptr;
            if (by_shape)
                snprintf(nbuf, sizeof(nbuf), "%s%d/k%d %dx%dx%d ks%d%s%s%s%s%s%s", pg ? "pg" : "v", var, sp, p.M, p.Cout, p.K, p.KS, res ? " res" : "",
                         geglu ? " geglu" : "", p.ln_stat ? " ln" : "", p.rowstat_out ? " rs" : "", p.gnstat_out ? " gs" : "",
                         p.gni_part ? " gn" : "");
            name = nbuf;
        }
        prof_open(c.stream, name, 2.0 * p.M * w.cout * kreal,
                  2.0 * (in_px * (w.ks == 1 ? (double)w.K : w.cin) + (double)w.cout * w.K +
                         (double)p.M * (geglu ? w.cout / 2 : w.cout) * (res ? 2 : 1)));
    }
    c.err = v2 ? launch_igemm2(p, partial, c.stream) : launch_igemm(p, c.stream);
    prof_close(c.stream);
    if (stat_out && !own_stats && !c.err) {
        prof_open(c.stream, "row_stats_kernel", 0.0, 2.0 * p.M * w.cout);
        c.err = launch_row_stats(y.p, y.ld, stat_out->p, p.M, w.cout, c.stream);
        prof_close(c.stream);
    }
}

void op_groupnorm(Ctx& c, const NormW& n, View x, View y, int N, long HW, int G, float eps, int silu,
                  const GnStatBuf* pre) {
    float* scratch = c.arena->alloc_f(gn_scratch_floats(N, HW, n.C, G));
    if (c.dry || c.err) return;
    static const bool gn_old = getenv("SD_GN_OLD") != nullptr;
    (void)gn_old;
    const GnStats* st = (pre && pre->st.part && gn_wants_stats(HW, n.C, G)) ? &pre->st : nullptr;
    // bytes really moved: the single-kernel form and the apply pass read x and write y; only the
    // stand-alone statistics pass reads x once more
    const bool own_pass = gn_wants_stats(HW, n.C, G) && !st;
    const char* gname = own_pass ? "groupnorm(stats+apply)" : "groupnorm(apply)";
    static thread_local char gbuf[64];
    static const bool gn_by_shape = getenv("SD_PROF_SHAPES") != nullptr;
    if (gn_by_shape && prof_enabled()) {
        snprintf(gbuf, sizeof(gbuf), "gn%s %ldx%d%s", own_pass ? "(stats+apply)" : (st ? "(apply)" : "(fused)"), (long)N * HW, n.C, silu ? " silu" : "");
        gname = gbuf;
    }
    prof_open(c.stream, gname, 0.0, (own_pass ? 6.0 : 4.0) * N * HW * n.C);
    c.err = launch_groupnorm(x.p, x.ld, n.gamma, n.beta, y.p, y.ld, N, HW, n.C, G, eps, silu, scratch, c.stream, st);
    prof_close(c.stream);
}

void op_gn_conv(Ctx& c, const NormW& n, const ConvW& w, View x, int N, int H, int W, View y, int G, float eps, int silu,
                const GnStatBuf* pre, const float* rowadd, int rowadd_ld, const View* res, const ConvFuse* fuse) {
    // Off unless SD_GN_FUSE=1: measured slower than GroupNorm kernel + convolution on every UNet shape (the in-LDS
    // transform does not hide behind the MFMAs, profiles/r03_gn_fused_conv.txt); the kernel stays for the operator
    // test (sd_op_groupnorm_conv2d) and as the base of the next attempt.
    static const bool off = getenv("SD_GN_FUSE") == nullptr || getenv("SD_NO_GN_FUSE") != nullptr;
    const long HW = (long)H * W;
    // the problem as op_conv will pose it (stride 1, no upsample: the only form that follows a GroupNorm)
    IGemmParams q;
    q.x = x.p; q.ldx = x.ld; q.N = N; q.H = H; q.W = W; q.Cin = (w.ks == 1) ? (int)w.K : w.cin;
    q.KS = w.ks; q.stride = 1; q.up = 0; q.pad = w.ks == 3 ? 1 : 0; q.OH = H; q.OW = W; q.Cout = w.cout;
    q.M = N * H * W; q.K = (int)w.K;
    const bool fused = !off && n.gb && n.C == q.Cin && igemm2_gn_fusable(q, G);
    if (!fused) {
        const size_t mk = c.arena->mark();
        View hn(c.arena->alloc_h((long)N * HW * n.C), n.C, n.C);
        op_groupnorm(c, n, x, hn, N, HW, G, eps, silu, pre);
        op_conv(c, w, hn, N, H, W, y, 1, 0, rowadd, rowadd_ld, res, 0, -1, 0, fuse);
        c.arena->release(mk);
        return;
    }
    const size_t mk = c.arena->mark();
    ConvFuse f = fuse ? *fuse : ConvFuse();
    f.gn_in = &n; f.gn_in_groups = G; f.gn_in_eps = eps; f.gn_in_silu = silu;
    float* scratch = c.arena->alloc_f(gn_scratch_floats(N, HW, n.C, G) + (long)N * G * 2);
    if (pre && pre->st.part) {
        f.gn_in_stats = pre->st;
        // many summaries per image (the VAE's 512 x 512 maps): merged once by a small kernel, not by every block's prologue
        if (f.gn_in_stats.S > 64 && !c.dry && !c.err) {
            prof_open(c.stream, "gn_finalize_kernel", 0.0, 8.0 * N * f.gn_in_stats.S * G);
            c.err = launch_gn_finalize(&f.gn_in_stats, scratch, N, HW, n.C, G, c.stream);
            prof_close(c.stream);
        }
    } else if (!c.dry && !c.err) {
        static thread_local char gbuf[64];
        static const bool by_shape = getenv("SD_PROF_SHAPES") != nullptr;
        const char* name = "groupnorm(stats)";
        if (by_shape && prof_enabled()) { snprintf(gbuf, sizeof(gbuf), "gn(stats) %ldx%d", (long)N * HW, n.C); name = gbuf; }
        prof_open(c.stream, name, 0.0, 2.0 * N * HW * n.C);
        c.err = launch_gn_stats(x.p, x.ld, N, HW, n.C, G, scratch, &f.gn_in_stats, c.stream);
        prof_close(c.stream);
    }
    op_conv(c, w, x, N, H, W, y, 1, 0, rowadd, rowadd_ld, res, 0, -1, 0, &f);
    c.arena->release(mk);
}

bool op_ffn_fused(Ctx& c, const ConvW& ff1, const ConvW& ff2, View x, const RowStat& x_stat, float ln_eps, long M, View y) {
    FfnParams p;
    p.x = x.p; p.ldx = x.ld; p.y = y.p; p.ldy = y.ld;
    p.w1 = ff1.w; p.b1 = ff1.bias; p.wsum1 = ff1.wsum;
    p.w1_rows = (int)(((long)ff1.cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad);
    p.w2 = ff2.w; p.b2 = ff2.bias;
    p.w2_rows = (int)(((long)ff2.cout + kWeightRowPad - 1) / kWeightRowPad * kWeightRowPad);
    p.ln_stat = x_stat.p; p.ln_parts = x_stat.parts; p.ln_eps = ln_eps;
    p.M = (int)M; p.C = (int)ff1.K; p.hidden = (int)ff2.K;
    if (c.dry) p.ln_stat = reinterpret_cast<const float*>(8);          // planning pass: only "is set" matters
    if (ff1.ks != 1 || ff2.ks != 1 || ff2.cout != ff1.K || ff1.cout != 2 * ff2.K || x.C != p.C || !ff1.wsum) return false;
    if (c.dry ? !(p.C == 320 && p.hidden == 1280 && M % 128 == 0 && M / 128 >= 64 && !getenv("SD_NO_FFN_FUSE")) : !ffn_fused_supported(p))
        return false;
    if (c.dry || c.err) return true;
    prof_open(c.stream, "ffn_fused_kernel", 2.0 * M * ((double)ff1.cout * ff1.K + (double)ff2.cout * ff2.K),
              2.0 * ((double)M * p.C * 2 + (double)ff1.cout * ff1.K + (double)ff2.cout * ff2.K));
    c.err = launch_ffn_fused(p, c.stream);
    prof_close(c.stream);
    return true;
}

void op_layernorm(Ctx& c, const NormW& n, View x, View y, long rows, float eps) {
    if (c.dry || c.err) return;
    prof_open(c.stream, "layernorm_kernel", 0.0, 4.0 * rows * n.C);
    c.err = launch_layernorm(x.p, x.ld, n.gamma, n.beta, y.p, y.ld, rows, n.C, eps, c.stream);
    prof_close(c.stream);
}

void op_attention(Ctx& c, View q, View k, View v, View out, int B, int Tq, int Tk, int heads, int d, int causal,
                  int prescaled) {
    if (c.dry || c.err) return;
    if (prof_enabled()) {
        static thread_local char name[32];
        snprintf(name, sizeof(name), "attn_kernel<%d>", d);
        prof_open(c.stream, name, 4.0 * B * heads * (double)Tq * Tk * d,
                  2.0 * B * heads * d * (2.0 * Tq + 2.0 * Tk));
    }
    c.err = launch_attention(q.p, k.p, v.p, out.p, B, Tq, Tk, heads, d, q.ld, k.ld, v.ld, out.ld, c.stream, causal, prescaled);
    prof_close(c.stream);
}

}  // namespace sd
