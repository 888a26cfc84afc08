// Host-side runtime of the denoise engine: device memory arena, packed-weight store and the
// execution context shared by the UNet and VAE graphs.  Plain C++ + HIP runtime; no torch.
#pragma once
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

namespace sd {

// Activation view: NHWC fp16 tensor living at `p` with row (pixel) stride `ld` elements.
struct View {
    half_t* p = nullptr;
    long ld = 0;
    int C = 0;
    View() {}
    View(half_t* p_, long ld_, int C_) : p(p_), ld(ld_), C(C_) {}
    View slice(int c0, int c) const { return View(p + c0, ld, c); }
};

// Bump allocator over one hipMalloc'ed slab with stack-style mark/release.  In "dry" mode no memory
// is attached and only the peak is recorded; the real slab is sized from that.
class Arena {
  public:
    ~Arena();
    int reserve(size_t bytes);            // (re)allocate the slab if smaller than `bytes`
    void begin(bool dry) { dry_ = dry; off_ = 0; if (dry) peak_ = 0; }
    void* alloc(size_t bytes);
    half_t* alloc_h(long n) { return static_cast<half_t*>(alloc((size_t)n * sizeof(half_t))); }
    float* alloc_f(long n) { return static_cast<float*>(alloc((size_t)n * sizeof(float))); }
    size_t mark() const { return off_; }
    void release(size_t m) { off_ = m; }
    size_t peak() const { return peak_; }
    size_t capacity() const { return cap_; }
    bool dry() const { return dry_; }
    bool overflow() const { return overflow_; }

  private:
    char* base_ = nullptr;
    size_t cap_ = 0, off_ = 0, peak_ = 0;
    bool dry_ = false, overflow_ = false;
};

struct ConvW {          // conv or linear, packed [rows_pad][K] fp16 + fp32 bias (padded)
    half_t* w = nullptr;
    float* bias = nullptr;
    int cin = 0, cout = 0, ks = 1;
    long K = 0;         // packed K (>= ks*ks*cin, multiple of 64)
    float* wsum = nullptr;   // row sums of the packed weights: set when a LayerNorm was folded in (fold_ln)
};
// Per-row (sum, sum of squares) partials of an activation tensor, `parts` pairs per row: written by
// the GEMM that produced the tensor (or by launch_row_stats), consumed by the GEMM that applies the
// folded LayerNorm (IGemmParams::rowstat_out / ln_stat).
struct RowStat {
    float* p = nullptr;
    int parts = 0;
};
// GroupNorm summaries of an activation tensor written by the convolution that produced it: `buf` is
// caller-allocated (gnstat_floats), `st` is filled in by op_conv when the launch really emits them
// (st.part == nullptr otherwise: the GroupNorm then runs its own statistics pass).
struct GnStatBuf {
    float* buf = nullptr;
    GnStats st;
};
// What an op_conv launch fuses besides bias / row add / residual / GEGLU / activation.
struct ConvFuse {
    const RowStat* ln_in = nullptr;     // LayerNorm of the INPUT, folded into this linear (ConvW::wsum set)
    float ln_eps = 0.f;
    RowStat* stat_out = nullptr;        // row statistics of the OUTPUT for the LayerNorm that follows
    GnStatBuf* gn_out = nullptr;        // GroupNorm summaries of the OUTPUT for the GroupNorm that follows
    int gn_groups = 0;
    // GroupNorm (+ SiLU) of the INPUT applied inside the convolution (set by op_gn_conv when igemm2_gn_fusable):
    const struct NormW* gn_in = nullptr;
    GnStats gn_in_stats;                // (mean, M2) summaries of the input
    int gn_in_groups = 0;
    float gn_in_eps = 0.f;
    int gn_in_silu = 0;
    // y = acc_scale * (x W^T) + bias_scale * bias + ...: the range-scaled VAE encoder (powers of two; VAE::run_encode)
    float acc_scale = 1.f, bias_scale = 1.f;
};
struct NormW {
    float* gamma = nullptr;
    float* beta = nullptr;
    float* gb = nullptr;     // [C / 64][64 gamma | 64 beta]: what the fused GroupNorm's DMA piece reads (C % 64 == 0)
    int C = 0;
};

// Raw tensors handed over through sd_*_set_weight, kept on device as fp16 until finalize().
struct RawTensor {
    std::vector<int64_t> shape;
    half_t* dev = nullptr;
    long numel = 0;
};

class WeightStore {
  public:
    ~WeightStore();
    void declare(const std::string& key, std::vector<int64_t> shape);
    int set(const std::string& key, const void* data, const int64_t* shape, int ndim, int dtype);
    bool complete(std::string* missing) const;
    const RawTensor* raw(const std::string& key) const;
    // packing helpers (device work on the null stream; finalize() synchronises once)
    int pack_conv(const std::string& prefix, ConvW* out, bool has_bias = true);
    int pack_rows(const std::vector<std::string>& weight_keys, const std::vector<std::string>& bias_keys,
                  ConvW* out);                                  // row-concatenated linears
    int pack_geglu(const std::string& prefix, ConvW* out);      // [8C][C] -> 64-row interleave
    int pack_norm(const std::string& prefix, NormW* out);
    // Fold LayerNorm `ln` (applied to the GEMM's input) into the packed linear `w`: W <- W diag(gamma)
    // (rows < rows_scaled additionally times row_scale), bias <- bias + W beta, w->wsum = row sums.
    int fold_ln(ConvW* w, const NormW& ln, int rows_scaled, float row_scale);
    void free_raw();
    void* dmalloc(size_t bytes);                                // tracked device allocation
    int64_t packed_bytes() const { return packed_bytes_; }

    std::vector<std::string> order;                             // manifest order
    std::unordered_map<std::string, RawTensor> tensors;

  private:
    int host_floats(const std::string& key, std::vector<float>* out) const;
    std::vector<void*> owned_;
    int64_t packed_bytes_ = 0;
};

// Execution context of one forward call.
struct Ctx {
    Arena* arena;
    hipStream_t stream;
    bool dry;
    int err = 0;
    // Ring of GroupNorm-summary buffers (ctx_gnpool_init at the top of a forward): a convolution whose
    // output goes into a GroupNorm takes the next slot (ctx_gnbuf) and the GroupNorm reads it one or two
    // launches later, long before the ring comes round again.
    static constexpr int kGnPool = 4;
    GnStatBuf gnpool[kGnPool];
    int gn_next = 0;
    int gn_groups = 0;
};
void ctx_gnpool_init(Ctx& c, int N, long HW_max, int G);
GnStatBuf* ctx_gnbuf(Ctx& c);       // next ring slot with `st` cleared; nullptr when the pool is not set up

// ---- optional per-launch timing (bench.py's live roofline): HIP events on the launch stream ----
struct ProfAgg { double flops = 0, bytes = 0, ms = 0; long launches = 0; };
void prof_enable(bool on);
bool prof_enabled();
void prof_open(hipStream_t s, const char* kernel, double flops, double bytes);
void prof_close(hipStream_t s);
int prof_collect(std::map<std::string, ProfAgg>* out);   // synchronises, aggregates by kernel name

// ---- op wrappers: skip the launch in dry mode, latch the first error ----
void op_conv(Ctx& c, const ConvW& w, View x, int N, int H, int W, View y, int stride = 1, int up = 0,
             const float* rowadd = nullptr, int rowadd_ld = 0, const View* res = nullptr, int geglu = 0,
             int pad = -1, int act = 0, const ConvFuse* fuse = nullptr);
// floats a RowStat buffer needs for an [M, C] tensor whatever tile the producer picks
inline long rowstat_floats(long M, int C) { return M * ((C + 63) / 64) * 2; }
// floats a GnStatBuf needs for N images of HW pixels (tiles of >= 64 pixels) and G groups
// (a split-K reduction kernel leaves up to 64 summaries per image whatever the map size)
inline long gnstat_floats(int N, long HW, int G) { const long t = (HW + 63) / 64; return (long)N * (t > 64 ? t : 64) * G * 2; }
void op_groupnorm(Ctx& c, const NormW& n, View x, View y, int N, long HW, int G, float eps, int silu,
                  const GnStatBuf* pre = nullptr);
// y = conv(act(GroupNorm(x))): inside the convolution when the launch can apply the norm to its halo tiles in LDS
// (igemm2_gn_fusable; the normalised tensor never exists), otherwise GroupNorm kernel + convolution.  `pre` = the
// summaries of x its producer left (or nullptr: a statistics pass runs); the other arguments as op_conv's.
void op_gn_conv(Ctx& c, const NormW& n, const ConvW& w, View x, int N, int H, int W, View y, int G, float eps, int silu,
                const GnStatBuf* pre, const float* rowadd = nullptr, int rowadd_ld = 0, const View* res = nullptr,
                const ConvFuse* fuse = nullptr);
void op_layernorm(Ctx& c, const NormW& n, View x, View y, long rows, float eps);
// y = x + GEGLU(LN(x) ff1) ff2 in one launch (ffn.hip) when the problem fits it (C = 320, LayerNorm folded into ff1, the
// row statistics of x at hand); returns false -- nothing launched -- when it does not: the caller runs the two GEMMs.
bool op_ffn_fused(Ctx& c, const ConvW& ff1, const ConvW& ff2, View x, const RowStat& x_stat, float ln_eps, long M, View y);
void op_attention(Ctx& c, View q, View k, View v, View out, int B, int Tq, int Tk, int heads, int d, int causal = 0,
                  int prescaled = 0);

}  // namespace sd
