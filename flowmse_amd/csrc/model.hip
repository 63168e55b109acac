// Model handle: NCSN++ module list, parameter table, weight packing and the static launch plan.
//
// Mirrors (reference) NCSNpp.__init__ (flowmse/backbones/ncsnpp.py:97-245: module order = parameter order)
// and NCSNpp.forward (:247-404: the order the modules are consumed in).  The forward pass is "traced" once per
// input shape into a flat list of kernel launches over an arena-planned activation workspace: no allocation,
// no host synchronisation and no shape logic inside the N-step solver loop.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/flowse_hip.h"
#include "common.h"

namespace flowse {

// ------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return ERR_HIP;
}

// ------------------------------------------------------------------------------------------- structure
enum ModKind { M_GFP, M_LINEAR, M_CONV3, M_RESBLOCK, M_ATTN, M_COMBINE, M_GN };

struct ParamInfo {
    std::string name;
    int ndim;
    int64_t shape[4];
    int64_t offset, numel;
};

struct Module {
    ModKind kind;
    int in_ch = 0, out_ch = 0;
    bool up = false, down = false, shortcut = false;
    int p0 = 0;            // index of the module's first parameter
    // offsets (floats) into the native device weight blob
    int64_t w_gn0_g = -1, w_gn0_b = -1, w_gn1_g = -1, w_gn1_b = -1;
    int64_t w_c0 = -1, w_c1 = -1, w_c1_b = -1, w_c2 = -1, w_c2_b = -1;
    int64_t w_a = -1, w_a_b = -1;      // generic weight / bias (linear, conv3, combine, gfp, gn gamma/beta)
    int64_t w_qkv = -1, w_qkv_b = -1, w_o = -1, w_o_b = -1;
    int dense_row0 = -1;               // first row of this block in the stacked Dense_0 table
    int64_t wq_c0 = -1, wq_c1 = -1;    // offsets (uint16 elements) of the bf16 planes of Conv_0 / Conv_1, if packed
};

struct Tn {
    size_t off = 0;
    int B = 0, H = 0, W = 0, C = 0;
    size_t st_off = 0;      // fused GroupNorm partials written by the producing conv (st_nblk blocks per sample)
    int st_nblk = 0;
    int dt = DT_F32;        // storage type of the elements (DT_*)
    size_t bytes() const { return (size_t)B * H * W * C * dt_size(dt); }
    bool valid() const { return B > 0; }
};

class Arena {
   public:
    size_t alloc(size_t n) {
        n = (n + 255) & ~(size_t)255;
        for (size_t i = 0; i < free_.size(); ++i) {
            if (free_[i].second >= n) {
                const size_t off = free_[i].first;
                if (free_[i].second == n) free_.erase(free_.begin() + i);
                else { free_[i].first += n; free_[i].second -= n; }
                live_[off] = n;
                return off;
            }
        }
        // extend (merge with a trailing free block if it touches the end)
        size_t off = end_;
        if (!free_.empty() && free_.back().first + free_.back().second == end_) {
            off = free_.back().first;
            free_.pop_back();
        }
        end_ = off + n;
        if (end_ > peak_) peak_ = end_;
        live_[off] = n;
        return off;
    }
    // Between a fork and its join the launch list runs on two streams: memory released by one branch must not be handed
    // to the other before the join, so releases are parked and applied at the join (plan-time bookkeeping only).
    void defer_releases(bool on) {
        defer_ = on;
        if (!on) {
            std::vector<size_t> d;
            d.swap(deferred_);
            for (size_t off : d) release(off);
        }
    }
    void release(size_t off) {
        if (defer_) {
            deferred_.push_back(off);
            return;
        }
        auto it = live_.find(off);
        if (it == live_.end()) return;
        const size_t n = it->second;
        live_.erase(it);
        size_t i = 0;
        while (i < free_.size() && free_[i].first < off) ++i;
        free_.insert(free_.begin() + i, std::make_pair(off, n));
        if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) {
            free_[i].second += free_[i + 1].second;
            free_.erase(free_.begin() + i + 1);
        }
        if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) {
            free_[i - 1].second += free_[i].second;
            free_.erase(free_.begin() + i);
        }
    }
    size_t peak() const { return peak_; }

   private:
    std::vector<std::pair<size_t, size_t>> free_;   // sorted by offset
    std::map<size_t, size_t> live_;
    size_t end_ = 0, peak_ = 0;
    bool defer_ = false;
    std::vector<size_t> deferred_;
};

struct Plan {
    int B = 0, F = 0, T = 0;
    size_t ws_bytes = 0;
    std::vector<std::function<int(hipStream_t)>> ops;
    std::vector<std::string> labels;
    std::vector<double> flops, bytes;      // algorithmic work / HBM traffic of each launch
    std::vector<double> issued;            // FLOPs the matrix cores execute for it (Winograd forms: 1/2 or 2/3 of `flops`)
    // Two-stream execution: ops flagged `side` run on the handle's side stream.  sync bit 0 (before the op): the side stream
    // waits for everything enqueued on the main stream so far ("fork"); bit 1: the main stream waits for the side stream
    // ("join").  Captured into the hipGraph as parallel branches; the NULL stream / profiling run everything in list order.
    std::vector<char> side, sync;
    std::vector<char> dominant;            // 1 = a launch of the dominant kernel: the unsplit 3x3 ResBlock conv with fused
                                           // GroupNorm+SiLU input (conv3x3_f43_kernel<2, false, 2> in the fp32 mode)
    // The launch list holds no per-call argument (those live in the handle's device-resident CallBlock), so after one
    // eager pass it is captured as a hipGraph and replayed: one graph launch per network evaluation.
    int eager_runs = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct ProfAcc {
    int64_t launches = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0, issued = 0.0;
};

}  // namespace flowse

using namespace flowse;

static constexpr int SK_TICKETS = 4096;   // split-K launches have < 256 tiles by policy; far above any grid.x they use

struct flowse_model {
    flowse_config cfg;
    std::vector<Module> mods;
    std::vector<ParamInfo> params;
    int64_t blob_numel = 0;
    int out_w_p = 0;                       // parameter index of output_layer.weight
    int temb_dim = 0, dense_rows = 0;
    int64_t w_dense = -1, w_dense_b = -1;  // stacked Dense_0 (+ folded Conv_0.bias)
    int64_t w_out = -1, w_out_b = -1;
    // device state
    float* d_w = nullptr;                  // native weight blob
    int64_t d_w_numel = 0;
    int precision = 0;                     // 0 fp32 (exact), 1 bf16x3 split (fp32-class), 2 bf16, 3 fp16 operands
    uint16_t* d_wq = nullptr;              // bf16 planes of the 3x3 ResBlock convs (precision != 0)
    int64_t d_wq_numel = 0;
    // 16-bit STORAGE modes (precision 2 / 3 on networks whose wide channel counts are multiples of 32): activations
    // between kernels are bf16 / half; d_w16 is an elementwise 16-bit copy of the packed weight blob d_w (same offsets)
    int act_dt = DT_F32;
    uint16_t* d_w16 = nullptr;
    int64_t d_w16_numel = 0;
    float* d_wino = nullptr;               // F(2,3) Winograd weights of the 3x3 convs the halo kernel can take
    int64_t d_wino_numel = 0;
    std::map<int64_t, int64_t> wino_of;    // packed weight offset (d_w) -> offset in d_wino
    char* d_ws = nullptr;                  // activation workspace
    size_t d_ws_bytes = 0;
    float* d_ts = nullptr;                 // [N][B] solver times
    size_t d_ts_floats = 0;
    std::map<std::tuple<int, int, int>, Plan> plans;
    CallBlock* d_call = nullptr;           // per-call arguments of the boundary kernels, in device memory
    unsigned* d_ticket = nullptr;          // split-K arrival counters, one per output tile (zero between launches)
    int device = -1;                       // HIP device that owns every d_* buffer of this handle
    bool use_graph = false;                // FLOWSE_GRAPH=1: replay each shape's launch list as a hipGraph (slower, measured)
    // Callers on the NULL (legacy default) stream -- PyTorch's default stream IS the NULL stream -- cannot be captured;
    // their work runs on this internal stream instead, fenced against the NULL stream by events on both sides.
    hipStream_t gstream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int64_t graph_launches = 0;            // hipGraphLaunch calls made by this handle (flowse_model_graph_launches)
    // side branch of the launch list (Plan::side): its own stream, fork / join events from a small pool
    hipStream_t sstream = nullptr;
    std::vector<hipEvent_t> br_events;
    bool use_branches = false;             // FLOWSE_BRANCH=1: the shortcut branch on a second stream (slower, measured)
    float* d_rk = nullptr;                 // fixed-step RK scratch: stage input + slope accumulator, 2 x [B,1,F,T] complex64
    size_t d_rk_floats = 0;
    // single-module handles (flowse_block_create): one ResnetBlockBigGANpp / AttnBlockpp / Combine behind the same
    // weight packer, plan builder and kernels as the full network -- unit parity against the reference's modules
    int block_kind = -1;                   // -1: full network; else FLOWSE_BLOCK_*
    struct BlockCall { const float* in1 = nullptr; const float* in2 = nullptr; const float* temb_act = nullptr;
                       float* out = nullptr; } bcall;
    std::map<std::tuple<int, int, int, int>, Plan> block_plans;      // (B, H, W, C1)
    // optional in-library profiler (flowse_profile_begin / _end): HIP events around selected launches
    int prof_mode = -1;                    // -1 off, 0 dominant kernel only, 1 every op
    std::vector<hipEvent_t> prof_pool;     // reusable events
    size_t prof_used = 0;
    struct Pending { int label; hipEvent_t a, b; double flops, bytes, issued; };
    std::vector<Pending> prof_pending;
    std::vector<std::string> prof_labels;
    std::map<std::string, int> prof_label_ix;
    double prof_tot_flops = 0.0, prof_tot_issued = 0.0;    // over every launch between _begin and _end
    int64_t prof_tot_launches = 0;

    float* W(int64_t off) const { return d_w + off; }
    float* A(size_t off) const { return reinterpret_cast<float*>(d_ws + off); }
    bool storage16() const { return act_dt != DT_F32; }
};

namespace flowse {

static bool in_list(const int32_t* v, int n, int x) {
    for (int i = 0; i < n; ++i)
        if (v[i] == x) return true;
    return false;
}

// 16-bit storage applies when every wide tensor of the network has a multiple of 32 channels (what the 16-bit
// matrix-core kernels tile by); otherwise precision 2 / 3 only switch the operands of the big 3x3 convs (fp32 storage).
static int storage_type_for(const flowse_model* m) {
    if (m->precision < 2 || getenv("FLOWSE_FP32_STORAGE")) return DT_F32;
    for (const auto& mod : m->mods) {
        if (mod.kind == M_RESBLOCK || mod.kind == M_ATTN || mod.kind == M_GN)
            if ((mod.in_ch % 32) || (mod.out_ch % 32)) return DT_F32;
        if ((mod.kind == M_COMBINE && (mod.out_ch % 32)) || (mod.kind == M_CONV3 && mod.out_ch != 4 && (mod.out_ch % 32)))
            return DT_F32;
    }
    return m->precision == 2 ? DT_BF16 : DT_F16;
}

// ---- parameter table helpers
static void add_param(flowse_model* m, const std::string& name, std::initializer_list<int64_t> shape) {
    ParamInfo p;
    p.name = name;
    p.ndim = (int)shape.size();
    p.numel = 1;
    int i = 0;
    for (int64_t s : shape) {
        p.shape[i++] = s;
        p.numel *= s;
    }
    for (; i < 4; ++i) p.shape[i] = 1;
    p.offset = m->blob_numel;
    m->blob_numel += p.numel;
    m->params.push_back(p);
}

static void add_module(flowse_model* m, Module mod) {
    const int idx = (int)m->mods.size();
    const std::string pre = "all_modules." + std::to_string(idx) + ".";
    mod.p0 = (int)m->params.size();
    const int64_t ci = mod.in_ch, co = mod.out_ch, td = m->temb_dim;
    switch (mod.kind) {
        case M_GFP:
            add_param(m, pre + "W", {co});
            break;
        case M_LINEAR:
            add_param(m, pre + "weight", {co, ci});
            add_param(m, pre + "bias", {co});
            break;
        case M_CONV3:
            add_param(m, pre + "weight", {co, ci, 3, 3});
            add_param(m, pre + "bias", {co});
            break;
        case M_GN:
            add_param(m, pre + "weight", {co});
            add_param(m, pre + "bias", {co});
            break;
        case M_COMBINE:
            add_param(m, pre + "Conv_0.weight", {co, ci, 1, 1});
            add_param(m, pre + "Conv_0.bias", {co});
            break;
        case M_RESBLOCK:
            add_param(m, pre + "GroupNorm_0.weight", {ci});
            add_param(m, pre + "GroupNorm_0.bias", {ci});
            add_param(m, pre + "Conv_0.weight", {co, ci, 3, 3});
            add_param(m, pre + "Conv_0.bias", {co});
            add_param(m, pre + "Dense_0.weight", {co, td});
            add_param(m, pre + "Dense_0.bias", {co});
            add_param(m, pre + "GroupNorm_1.weight", {co});
            add_param(m, pre + "GroupNorm_1.bias", {co});
            add_param(m, pre + "Conv_1.weight", {co, co, 3, 3});
            add_param(m, pre + "Conv_1.bias", {co});
            if (mod.shortcut) {
                add_param(m, pre + "Conv_2.weight", {co, ci, 1, 1});
                add_param(m, pre + "Conv_2.bias", {co});
            }
            break;
        case M_ATTN:
            add_param(m, pre + "GroupNorm_0.weight", {co});
            add_param(m, pre + "GroupNorm_0.bias", {co});
            for (int k = 0; k < 4; ++k) {
                add_param(m, pre + "NIN_" + std::to_string(k) + ".W", {co, co});
                add_param(m, pre + "NIN_" + std::to_string(k) + ".b", {co});
            }
            break;
    }
    m->mods.push_back(mod);
}

static Module resblock(int in_ch, int out_ch, bool up = false, bool down = false) {
    Module r;
    r.kind = M_RESBLOCK;
    r.in_ch = in_ch;
    r.out_ch = out_ch;
    r.up = up;
    r.down = down;
    r.shortcut = (in_ch != out_ch) || up || down;     // layerspp.py:234-235
    return r;
}
static Module simple(ModKind k, int in_ch, int out_ch) {
    Module r;
    r.kind = k;
    r.in_ch = in_ch;
    r.out_ch = out_ch;
    return r;
}

// NCSNpp.__init__, ncsnpp.py:97-245
static int build_structure(flowse_model* m) {
    const flowse_config& c = m->cfg;
    if (c.nf < 4 || (c.nf & 3) || c.num_levels < 1 || c.num_levels > FLOWSE_MAX_LEVELS || c.num_res_blocks < 1 ||
        c.num_attn < 0 || c.num_attn > FLOWSE_MAX_ATTN || c.image_size < (1 << (c.num_levels - 1))) {
        set_error("invalid config: nf=%d levels=%d res_blocks=%d attn=%d image_size=%d", c.nf, c.num_levels,
                  c.num_res_blocks, c.num_attn, c.image_size);
        return ERR_ARG;
    }
    for (int i = 0; i < c.num_levels; ++i)
        if (c.ch_mult[i] < 1) {
            set_error("invalid ch_mult[%d]=%d", i, c.ch_mult[i]);
            return ERR_ARG;
        }
    const int nf = c.nf, L = c.num_levels;
    m->temb_dim = 4 * nf;
    // output_layer is registered before all_modules (ncsnpp.py:97) -> first in parameters()
    m->out_w_p = (int)m->params.size();
    add_param(m, "output_layer.weight", {2, 4, 1, 1});
    add_param(m, "output_layer.bias", {2});

    add_module(m, simple(M_GFP, 0, nf));
    add_module(m, simple(M_LINEAR, 2 * nf, 4 * nf));
    add_module(m, simple(M_LINEAR, 4 * nf, 4 * nf));
    add_module(m, simple(M_CONV3, 4, nf));
    std::vector<int> hs_c{nf};
    int in_ch = nf;
    for (int lv = 0; lv < L; ++lv) {
        const int res = c.image_size >> lv;
        for (int b = 0; b < c.num_res_blocks; ++b) {
            const int out_ch = nf * c.ch_mult[lv];
            add_module(m, resblock(in_ch, out_ch));
            in_ch = out_ch;
            if (in_list(c.attn_resolutions, c.num_attn, res)) add_module(m, simple(M_ATTN, in_ch, in_ch));
            hs_c.push_back(in_ch);
        }
        if (lv != L - 1) {
            add_module(m, resblock(in_ch, in_ch, false, true));
            add_module(m, simple(M_COMBINE, 4, in_ch));
            hs_c.push_back(in_ch);
        }
    }
    in_ch = hs_c.back();
    add_module(m, resblock(in_ch, in_ch));
    add_module(m, simple(M_ATTN, in_ch, in_ch));
    add_module(m, resblock(in_ch, in_ch));
    for (int lv = L - 1; lv >= 0; --lv) {
        const int res = c.image_size >> lv;
        for (int b = 0; b < c.num_res_blocks + 1; ++b) {
            const int out_ch = nf * c.ch_mult[lv];
            add_module(m, resblock(in_ch + hs_c.back(), out_ch));
            hs_c.pop_back();
            in_ch = out_ch;
        }
        if (in_list(c.attn_resolutions, c.num_attn, res)) add_module(m, simple(M_ATTN, in_ch, in_ch));
        add_module(m, simple(M_GN, in_ch, in_ch));
        add_module(m, simple(M_CONV3, in_ch, 4));
        if (lv != 0) add_module(m, resblock(in_ch, in_ch, true, false));
    }
    if (!hs_c.empty()) {
        set_error("internal: skip stack not empty");
        return ERR_STATE;
    }
    return OK;
}

// ------------------------------------------------------------------------------------------- weight packing
struct Packer {
    std::vector<float> host;
    struct WinoReq { int64_t off; int Cout, Cin; };
    std::vector<WinoReq> wino;             // 3x3 convs that also get F(2,3) weights (transformed on the device)
    int64_t put(int64_t n) {
        const int64_t off = ((int64_t)host.size() + 63) & ~(int64_t)63;
        host.resize(off + n, 0.f);
        return off;
    }
};

// conv weight [Cout][Cin][kh][kw] -> [Cout][kh*kw][Cin]
static int64_t pack_conv(Packer& pk, const float* src, int Cout, int Cin, int taps) {
    const int64_t off = pk.put((int64_t)Cout * taps * Cin);
    float* dst = pk.host.data() + off;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < taps; ++t)
                dst[((int64_t)co * taps + t) * Cin + ci] = src[((int64_t)co * Cin + ci) * taps + t];
    if (taps == 9 && (Cin % 32) == 0 && (Cout % 64) == 0) pk.wino.push_back({off, Cout, Cin});
    return off;
}
static int64_t pack_copy(Packer& pk, const float* src, int64_t n) {
    const int64_t off = pk.put(n);
    memcpy(pk.host.data() + off, src, n * sizeof(float));
    return off;
}

static int pack_weights(flowse_model* m, const float* blob, Packer& pk) {
    auto P = [&](int idx) { return blob + m->params[idx].offset; };
    // count Dense_0 rows
    int rows = 0;
    for (auto& mod : m->mods)
        if (mod.kind == M_RESBLOCK) {
            mod.dense_row0 = rows;
            rows += mod.out_ch;
        }
    m->dense_rows = rows;
    const int td = m->temb_dim;
    m->w_dense = pk.put((int64_t)rows * td);
    m->w_dense_b = pk.put(rows);
    if (m->block_kind < 0) {
        m->w_out = pack_copy(pk, P(m->out_w_p), 8);
        m->w_out_b = pack_copy(pk, P(m->out_w_p + 1), 2);
    }
    for (auto& mod : m->mods) {
        const int p = mod.p0, ci = mod.in_ch, co = mod.out_ch;
        switch (mod.kind) {
            case M_GFP:
                mod.w_a = pack_copy(pk, P(p), co);
                break;
            case M_LINEAR:
                mod.w_a = pack_copy(pk, P(p), (int64_t)co * ci);
                mod.w_a_b = pack_copy(pk, P(p + 1), co);
                break;
            case M_CONV3:
                mod.w_a = pack_conv(pk, P(p), co, ci, 9);
                mod.w_a_b = pack_copy(pk, P(p + 1), co);
                break;
            case M_GN:
                mod.w_a = pack_copy(pk, P(p), co);
                mod.w_a_b = pack_copy(pk, P(p + 1), co);
                break;
            case M_COMBINE:
                mod.w_a = pack_conv(pk, P(p), co, ci, 1);
                mod.w_a_b = pack_copy(pk, P(p + 1), co);
                break;
            case M_RESBLOCK: {
                mod.w_gn0_g = pack_copy(pk, P(p), ci);
                mod.w_gn0_b = pack_copy(pk, P(p + 1), ci);
                mod.w_c0 = pack_conv(pk, P(p + 2), co, ci, 9);
                // Dense_0 rows into the stacked table; Conv_0.bias folded into the table's bias
                memcpy(pk.host.data() + m->w_dense + (int64_t)mod.dense_row0 * td, P(p + 4),
                       (size_t)co * td * sizeof(float));
                for (int r = 0; r < co; ++r)
                    pk.host[m->w_dense_b + mod.dense_row0 + r] = P(p + 5)[r] + P(p + 3)[r];
                mod.w_gn1_g = pack_copy(pk, P(p + 6), co);
                mod.w_gn1_b = pack_copy(pk, P(p + 7), co);
                mod.w_c1 = pack_conv(pk, P(p + 8), co, co, 9);
                mod.w_c1_b = pack_copy(pk, P(p + 9), co);
                if (mod.shortcut) {
                    mod.w_c2 = pack_conv(pk, P(p + 10), co, ci, 1);
                    mod.w_c2_b = pack_copy(pk, P(p + 11), co);
                }
                break;
            }
            case M_ATTN: {
                const int C = co;
                mod.w_gn0_g = pack_copy(pk, P(p), C);
                mod.w_gn0_b = pack_copy(pk, P(p + 1), C);
                // NIN W is [in][out] (layers.py:549): transpose to [out][in]; q,k,v stacked -> [3C][C]
                mod.w_qkv = pk.put((int64_t)3 * C * C);
                mod.w_qkv_b = pk.put(3 * C);
                for (int k = 0; k < 3; ++k) {
                    const float* Wk = P(p + 2 + 2 * k);
                    const float* bk = P(p + 3 + 2 * k);
                    for (int o = 0; o < C; ++o) {
                        for (int i = 0; i < C; ++i)
                            pk.host[mod.w_qkv + ((int64_t)k * C + o) * C + i] = Wk[(int64_t)i * C + o];
                        pk.host[mod.w_qkv_b + k * C + o] = bk[o];
                    }
                }
                mod.w_o = pk.put((int64_t)C * C);
                const float* W3 = P(p + 8);
                for (int o = 0; o < C; ++o)
                    for (int i = 0; i < C; ++i) pk.host[mod.w_o + (int64_t)o * C + i] = W3[(int64_t)i * C + o];
                mod.w_o_b = pack_copy(pk, P(p + 9), C);
                break;
            }
        }
    }
    pk.host.resize((pk.host.size() + 63) & ~(size_t)63, 0.f);      // whole float4s (the 16-bit twin converts by quads)
    return OK;
}

// ------------------------------------------------------------------------------------------- plan builder
struct GnBuf {
    size_t mean = 0, scale = 0;
    int64_t beta = -1;
};

// A split-K convolution whose reduction is left to another conv's reduction launch (ConvArgs::partial2)
struct SkPartial {
    size_t part_off = 0;
    int ks = 0;
    int64_t bias = -1;
    bool valid = false;
};
// Request to fuse the GroupNorm that consumes a split-K conv's output into its reduction launch
// (launch_splitk_reduce_gn).  apply: the conv returns act(GroupNorm(out)); else it returns out and fills `g`.
struct GnFuse {
    int64_t w_gamma = -1, w_beta = -1;
    bool silu = true, apply = false;
    GnBuf g;
    bool done = false;
};

struct Builder {
    flowse_model* m;
    Plan* plan;
    Arena arena;
    int B;

    // dt < 0: the model's activation type for wide tensors, fp32 for the 4-channel ones (input pack, pyramids)
    Tn alloc(int H, int W, int C, int dt = -1) {
        Tn t;
        t.B = B; t.H = H; t.W = W; t.C = C;
        t.dt = dt >= 0 ? dt : (C > 4 ? m->act_dt : DT_F32);
        t.off = arena.alloc(t.bytes());
        return t;
    }
    void release(const Tn& t) {
        if (!t.valid()) return;
        arena.release(t.off);
        if (t.st_nblk > 0) arena.release(t.st_off);
    }
    void drop_stats(Tn& t) {          // the tensor was modified in place: its fused statistics are stale
        if (t.st_nblk > 0) arena.release(t.st_off);
        t.st_nblk = 0;
    }
    // fork(): the ops recorded until side_end() form a side branch that may run concurrently with the main-stream ops
    // recorded after it, up to join().  Usage: fork(); <side ops>; side_end(); <main ops>; join(); <consumer of both>.
    bool failed = false;                   // an internal planning inconsistency: build_plan returns ERR_STATE
    bool side_mode = false;
    char pending_sync = 0;
    bool branches = true;
    void fork() {
        if (!branches) return;
        side_mode = true;
        pending_sync |= 1;
        arena.defer_releases(true);
    }
    void side_end() { side_mode = false; }
    void join() {
        if (!branches) return;
        side_mode = false;
        pending_sync |= 2;
        arena.defer_releases(false);
    }
    void op(const std::string& label, std::function<int(hipStream_t)> f, double flops = 0.0, double bytes = 0.0,
            bool dominant = false, double issued = -1.0) {
        plan->side.push_back(side_mode ? 1 : 0);
        plan->sync.push_back(pending_sync);
        pending_sync = 0;
        plan->ops.push_back(std::move(f));
        plan->labels.push_back(label);
        plan->flops.push_back(flops);
        plan->bytes.push_back(bytes);
        plan->issued.push_back(issued < 0.0 ? flops : issued);
        plan->dominant.push_back(dominant ? 1 : 0);
    }

    // statistics (fused partials of the producing conv when present, else a gn_stats pass per tensor) +
    // finalize; returns per-(b,c) mean / scale buffers (caller releases)
    GnBuf gn(const Tn& a, const Tn* b2, int64_t w_gamma, int64_t w_beta) {
        flowse_model* M = m;
        const int C1 = a.C, C2 = b2 ? b2->C : 0, C = C1 + C2, HW = a.H * a.W, Bn = B;
        const int G = std::min(C / 4, 32);
        size_t poff[2] = {0, 0};
        int pnblk[2] = {0, 0};
        bool temp[2] = {false, false};
        const Tn* src[2] = {&a, b2};
        for (int k = 0; k < 2; ++k) {
            if (!src[k]) continue;
            if (src[k]->st_nblk > 0) {
                poff[k] = src[k]->st_off;
                pnblk[k] = src[k]->st_nblk;
                continue;
            }
            const int Ck = src[k]->C;
            const int nblk = gn_partial_blocks(HW, Ck);
            poff[k] = arena.alloc((size_t)Bn * nblk * Ck * 2 * sizeof(float));
            pnblk[k] = nblk;
            temp[k] = true;
            const size_t t_off = src[k]->off, p_off = poff[k];
            const int sdt = src[k]->dt;
            op("gn_stats@" + std::to_string(src[k]->H) + "x" + std::to_string(src[k]->W), [=](hipStream_t s) {
                return launch_gn_stats(M->A(t_off), Ck, nullptr, 0, Bn, HW, M->A(p_off), nblk, s, sdt);
            }, 3.0 * Bn * HW * Ck, (double)dt_size(sdt) * Bn * HW * Ck);
        }
        GnBuf g;
        g.mean = arena.alloc((size_t)Bn * C * sizeof(float));
        g.scale = arena.alloc((size_t)Bn * C * sizeof(float));
        g.beta = w_beta;
        const size_t gm = g.mean, gs = g.scale, p0 = poff[0], p1 = poff[1];
        const int n0 = pnblk[0], n1 = pnblk[1];
        const bool has2 = b2 != nullptr;
        op("gn_finalize@" + std::to_string(a.H) + "x" + std::to_string(a.W), [=](hipStream_t s) {
            return launch_gn_finalize(M->A(p0), n0, C1, has2 ? M->A(p1) : nullptr, n1, C2, Bn, HW, G, M->W(w_gamma),
                                      1e-6f, M->A(gm), M->A(gs), s);
        });
        for (int k = 0; k < 2; ++k)
            if (temp[k]) arena.release(poff[k]);
        return g;
    }
    // GroupNorm (+ SiLU) materialised into a new tensor.  Small images: statistics finalize and the apply pass are ONE
    // launch (a block per (group, sample) reduces the partials and normalises its HW x C/G elements); otherwise
    // finalize + float4 apply.
    // out_dt < 0: same storage type as the input
    Tn gn_norm(const Tn& a, const Tn* b2, int64_t w_gamma, int64_t w_beta, bool silu, int out_dt = -1) {
        const int C1 = a.C, C2 = b2 ? b2->C : 0, C = C1 + C2, HW = a.H * a.W, Bn = B;
        const int G = std::min(C / 4, 32);
        const int idt = a.dt, odt = out_dt >= 0 ? out_dt : a.dt;
        if ((int64_t)HW * (C / G) > 8192) {
            GnBuf g = gn(a, b2, w_gamma, w_beta);
            Tn o = gn_apply(a, b2, g, silu, odt);
            gn_release(g);
            return o;
        }
        flowse_model* M = m;
        size_t poff[2] = {0, 0};
        int pnblk[2] = {0, 0};
        bool temp[2] = {false, false};
        const Tn* src[2] = {&a, b2};
        for (int k = 0; k < 2; ++k) {
            if (!src[k]) continue;
            if (src[k]->st_nblk > 0) {
                poff[k] = src[k]->st_off;
                pnblk[k] = src[k]->st_nblk;
                continue;
            }
            const int Ck = src[k]->C;
            const int nblk = gn_partial_blocks(HW, Ck);
            poff[k] = arena.alloc((size_t)Bn * nblk * Ck * 2 * sizeof(float));
            pnblk[k] = nblk;
            temp[k] = true;
            const size_t t_off = src[k]->off, p_off = poff[k];
            op("gn_stats@" + std::to_string(src[k]->H) + "x" + std::to_string(src[k]->W), [=](hipStream_t s) {
                return launch_gn_stats(M->A(t_off), Ck, nullptr, 0, Bn, HW, M->A(p_off), nblk, s, idt);
            }, 3.0 * Bn * HW * Ck, (double)dt_size(idt) * Bn * HW * Ck);
        }
        Tn o = alloc(a.H, a.W, C, odt);
        const size_t a_off = a.off, b_off = b2 ? b2->off : 0, o_off = o.off, p0 = poff[0], p1 = poff[1];
        const int n0 = pnblk[0], n1 = pnblk[1];
        const bool has2 = b2 != nullptr;
        op("gn_norm@" + std::to_string(a.H) + "x" + std::to_string(a.W), [=](hipStream_t s) {
            return launch_gn_finalize_apply(M->A(a_off), M->A(p0), n0, C1, has2 ? M->A(b_off) : nullptr,
                                            has2 ? M->A(p1) : nullptr, n1, C2, Bn, HW, G, M->W(w_gamma), M->W(w_beta),
                                            1e-6f, silu ? 1 : 0, M->A(o_off), s, idt, odt);
        }, 8.0 * Bn * HW * C, (double)(dt_size(idt) + dt_size(odt)) * Bn * HW * C);
        for (int k = 0; k < 2; ++k)
            if (temp[k]) arena.release(poff[k]);
        return o;
    }
    void gn_release(const GnBuf& g) {
        arena.release(g.mean);
        arena.release(g.scale);
    }
    Tn gn_apply(const Tn& a, const Tn* b2, const GnBuf& g, bool silu, int out_dt = -1) {
        flowse_model* M = m;
        const int C1 = a.C, C2 = b2 ? b2->C : 0, HW = a.H * a.W, Bn = B;
        const int idt = a.dt, odt = out_dt >= 0 ? out_dt : a.dt;
        Tn o = alloc(a.H, a.W, C1 + C2, odt);
        const size_t a_off = a.off, b_off = b2 ? b2->off : 0, o_off = o.off;
        const bool has2 = b2 != nullptr;
        op("gn_apply@" + std::to_string(a.H) + "x" + std::to_string(a.W), [=](hipStream_t s) {
            GnParams p{M->A(g.mean), M->A(g.scale), M->W(g.beta)};
            return launch_gn_apply(M->A(a_off), C1, has2 ? M->A(b_off) : nullptr, C2, Bn, HW, p, silu ? 1 : 0,
                                   M->A(o_off), s, idt, odt);
        }, 8.0 * Bn * HW * (C1 + C2), (double)(dt_size(idt) + dt_size(odt)) * Bn * HW * (C1 + C2));
        return o;
    }
    // conv: out (new tensor unless `inplace_res`), res optional
    Tn conv(const std::string& label, const Tn& a, const Tn* b2, int64_t w, int64_t bias, int dense_row0, int Cout,
            int taps, const Tn* res, float scale, bool out_is_res = false, bool cin4 = false,
            const GnBuf* gin = nullptr, bool gin_silu = false, int64_t wq_off = -1, int out_dt = -1,
            SkPartial* defer = nullptr, const SkPartial* extra = nullptr, GnFuse* gnf = nullptr) {
        flowse_model* M = m;
        const int C1 = a.C, C2 = b2 ? b2->C : 0, H = a.H, Wd = a.W, Bn = B;
        {   // a deferred reduction leaves no output tensor: decide before anything is allocated
            const bool in16_ = a.dt != DT_F32;
            const int ks_ = cin4 ? 1 : in16_ ? ((conv_supports_head4(Bn, H, Wd, C1, C2, Cout, taps) || conv16_uses_halo(Bn, H, Wd, C1, C2, Cout, taps))
                                                    ? 1 : conv16_ksplit(Bn, H, Wd, C1 + C2, Cout, taps))
                                             : conv_ksplit(Bn, H, Wd, C1 + C2, Cout, taps);
            if (defer && !(ks_ > 1 && !conv_splitk_in_launch() && !res && dense_row0 < 0)) defer = nullptr;
            if ((extra || gnf) && !(ks_ > 1 && !conv_splitk_in_launch())) {
                if (extra && extra->valid) {
                    set_error("internal: merged reduction requested for an unsplit conv (%s)", label.c_str());
                    failed = true;
                }
                gnf = nullptr;
            }
            if (gnf && (res || !conv_reduce_gn_ok(Bn, H * Wd, Cout))) gnf = nullptr;
        }
        Tn o;
        if (!defer) o = out_is_res ? *res : alloc(a.H, a.W, Cout, out_dt);
        const size_t a_off = a.off, b_off = b2 ? b2->off : 0, o_off = o.off, r_off = res ? res->off : 0;
        const bool has2 = b2 != nullptr, hasres = res != nullptr;
        const int idt = a.dt, odt = o.dt;
        const bool in16 = idt != DT_F32;                 // 16-bit storage: the 16-bit matrix-core kernels take it
        const int ks = cin4 ? 1 : in16 ? (conv_supports_head4(Bn, H, Wd, C1, C2, Cout, taps) || conv16_uses_halo(Bn, H, Wd, C1, C2, Cout, taps)
                                              ? 1 : conv16_ksplit(Bn, H, Wd, C1 + C2, Cout, taps))
                                       : conv_ksplit(Bn, H, Wd, C1 + C2, Cout, taps);
        int st_nblk = (cin4 || out_is_res || Cout < 16) ? 0
                      : in16 ? (conv16_uses_halo(Bn, H, Wd, C1, C2, Cout, taps) ? H * Wd / 128
                                                                                 : conv16_stats_blocks(Bn, H, Wd, C1 + C2, Cout, taps))
                             : conv_fused_stats_blocks(Bn, H, Wd, C1 + C2, Cout, taps);
        if (cin4 && !out_is_res && C1 == 4 && !has2 && conv_cin4_uses_mfma(Bn, H, Wd, Cout, taps)) st_nblk = H * Wd / 128;
        if (defer || gnf) st_nblk = 0;                   // no output here / the statistics are finished inside the reduction
        if (st_nblk > 0) {
            o.st_nblk = st_nblk;
            o.st_off = arena.alloc((size_t)Bn * st_nblk * Cout * 2 * sizeof(float));
        }
        const size_t st_off = o.st_off;
        const bool has_gin = gin != nullptr;
        const GnBuf gbuf = has_gin ? *gin : GnBuf();
        const bool use_bf16 = !M->storage16() && wq_off >= 0 && M->precision != 0 && taps == 9 &&
                              conv_supports_bf16(Bn, H, Wd, C1, C2, Cout, taps);
        const int terms = M->precision == 1 ? 3 : 1;
        const auto wino_it = M->wino_of.find(w);
        const int64_t wino_off = (!in16 && taps == 9 && !cin4 && wino_it != M->wino_of.end() &&
                                  conv_supports_wino(Bn, H, Wd, C1, C2, Cout, taps)) ? wino_it->second : -1;
        const size_t part_off = ks > 1 ? arena.alloc((size_t)ks * Bn * H * Wd * Cout * sizeof(float)) : 0;
        // split-K: reduce inside the launch (last-arriving slice, ConvArgs::sk_ticket) unless switched off
        const int64_t sk_tiles = (((int64_t)Bn * H * Wd + 127) / 128) * ((Cout + 31) / 32);
        const bool sk_in_launch = ks > 1 && conv_splitk_in_launch() && sk_tiles <= SK_TICKETS;
        const int sk_group = st_nblk > 0 ? H * Wd / st_nblk : 0;
        const size_t table_off = M_table_off;     // by value: the Builder dies before the plan runs
        const bool has_extra = extra != nullptr && extra->valid;
        const SkPartial xp = has_extra ? *extra : SkPartial();
        const std::string full_label = label + "@" + std::to_string(H) + "x" + std::to_string(Wd) + ":" +
                                       std::to_string(C1 + C2) + ">" + std::to_string(Cout);
        auto make_args = [=]() {
            ConvArgs c;
            c.in1 = M->A(a_off);
            c.in2 = has2 ? M->A(b_off) : nullptr;
            c.C1 = C1;
            c.C2 = C2;
            c.w = M->W(w);
            c.bias = bias >= 0 ? M->W(bias) : nullptr;
            c.bias2 = dense_row0 >= 0 ? M->A(table_off) + dense_row0 : nullptr;
            c.bias2_stride = M->dense_rows;
            c.res = hasres ? M->A(r_off) : nullptr;
            c.out = M->A(o_off);
            c.B = Bn; c.H = H; c.W = Wd; c.Cout = Cout;
            c.taps = taps;
            c.scale = scale;
            c.ksplit = ks;
            c.partial = ks > 1 ? M->A(part_off) : nullptr;
            if (has_extra) {
                c.partial2 = M->A(xp.part_off);
                c.ksplit2 = xp.ks;
                c.bias_x = xp.bias >= 0 ? M->W(xp.bias) : nullptr;
            }
            if (sk_in_launch) {
                c.sk_ticket = M->d_ticket;
                c.sk_group = sk_group;
            }
            c.stats = st_nblk > 0 ? M->A(st_off) : nullptr;
            c.stats_nblk = st_nblk;
            if (has_gin) {
                c.gn = GnParams{M->A(gbuf.mean), M->A(gbuf.scale), M->W(gbuf.beta)};
                c.gn_silu = gin_silu ? 1 : 0;
            }
            c.in_dt = idt;
            c.out_dt = odt;
            if (in16) {                                   // [Cout][taps][Cin] in the storage type: same offsets as d_w
                c.wq = M->d_w16 + w;
                c.terms = 1;
                c.wq_f16 = idt == DT_F16 ? 1 : 0;
            } else if (use_bf16) {
                c.wq = M->d_wq + wq_off;
                c.terms = terms;
                c.wq_f16 = M->precision == 3 ? 1 : 0;
            }
            if (wino_off >= 0) {
                c.wino = M->d_wino + wino_off;
                c.wino_f43 = conv_wino_default_f43() ? 1 : 0;
            }
            return c;
        };
        const double flops = 2.0 * Bn * H * Wd * (double)Cout * taps * (C1 + C2);
        const double out_bytes = (double)dt_size(odt) * Bn * H * Wd * Cout * (hasres ? 2 : 1);
        const double in_bytes = (double)dt_size(idt) * ((double)Bn * H * Wd * (C1 + C2) + (double)Cout * taps * (C1 + C2));
        const double part_bytes = 4.0 * ks * (double)Bn * H * Wd * Cout;
        op(full_label, [=](hipStream_t s) {
            const ConvArgs c = make_args();
            return cin4 ? launch_conv_cin4(c, s) : launch_conv(c, s, false);
        }, flops, in_bytes + (ks > 1 ? part_bytes + (sk_in_launch ? part_bytes + out_bytes : 0.0) : out_bytes),
           has_gin && Cout > 64 && ks == 1 &&
               (wino_off < 0 || !conv_wino_default_f43() || conv_f43_forced_bn64() || conv_f43_wide(Bn, H, Wd, Cout)),
           wino_off >= 0 ? flops * (conv_wino_default_f43() ? 0.5 : 2.0 / 3.0) : (use_bf16 && terms == 3) ? 3.0 * flops : flops);
        if (defer) {                                     // the consumer's reduction sums these slices (ConvArgs::partial2)
            defer->part_off = part_off;
            defer->ks = ks;
            defer->bias = bias;
            defer->valid = true;
            return Tn();
        }
        if (ks > 1 && !sk_in_launch && gnf) {
            const int64_t wg = gnf->w_gamma, wb = gnf->w_beta;
            const bool gsilu = gnf->silu, gapply = gnf->apply;
            GnBuf g;
            if (!gapply) {
                g.mean = arena.alloc((size_t)Bn * Cout * sizeof(float));
                g.scale = arena.alloc((size_t)Bn * Cout * sizeof(float));
                g.beta = wb;
            }
            const size_t gm = g.mean, gs = g.scale;
            op("splitk_reduce_gn@" + std::to_string(H) + "x" + std::to_string(Wd), [=](hipStream_t s) {
                return launch_splitk_reduce_gn(make_args(), M->W(wg), M->W(wb), 1e-6f, gsilu ? 1 : 0, gapply ? 1 : 0,
                                               gapply ? nullptr : M->A(gm), gapply ? nullptr : M->A(gs), s);
            }, 8.0 * Bn * H * Wd * Cout, part_bytes + out_bytes);
            gnf->g = g;
            gnf->done = true;
        } else if (ks > 1 && !sk_in_launch)
            op("splitk_reduce@" + std::to_string(H) + "x" + std::to_string(Wd), [=](hipStream_t s) { return launch_splitk_reduce(make_args(), s); }, 0.0,
               part_bytes * (has_extra ? 1.0 + (double)xp.ks / ks : 1.0) + out_bytes);
        if (ks > 1) arena.release(part_off);
        return o;
    }
    size_t M_table_off = 0;     // arena offset of the Dense_0 bias table [B][dense_rows]

    // raw_out (optional): receives the same resampling of the un-normalised input (one read of `a` for both)
    Tn fir(const Tn& a, bool up, const GnBuf* g, bool silu, const Tn* add, bool out_is_add = false,
           Tn* raw_out = nullptr) {
        flowse_model* M = m;
        const int H = a.H, Wd = a.W, C = a.C, Bn = B;
        const int fdt = a.dt;
        Tn o = out_is_add ? *add : (up ? alloc(2 * H, 2 * Wd, C, fdt) : alloc(H / 2, Wd / 2, C, fdt));
        if (raw_out) *raw_out = up ? alloc(2 * H, 2 * Wd, C, fdt) : alloc(H / 2, Wd / 2, C, fdt);
        const size_t a_off = a.off, o_off = o.off, add_off = add ? add->off : 0, r_off = raw_out ? raw_out->off : 0;
        const bool hasg = g != nullptr, hasadd = add != nullptr, hasraw = raw_out != nullptr;
        GnBuf gb = hasg ? *g : GnBuf();
        const double outs = hasraw ? 2.0 : 1.0;
        op(std::string(up ? "fir_up@" : "fir_down@") + std::to_string(H) + "x" + std::to_string(Wd), [=](hipStream_t s) {
            GnParams p{nullptr, nullptr, nullptr};
            if (hasg) p = GnParams{M->A(gb.mean), M->A(gb.scale), M->W(gb.beta)};
            if (up)
                return launch_fir_up(M->A(a_off), Bn, H, Wd, C, p, silu ? 1 : 0, hasadd ? M->A(add_off) : nullptr,
                                     M->A(o_off), s, hasraw ? M->A(r_off) : nullptr, fdt);
            return launch_fir_down(M->A(a_off), Bn, H, Wd, C, p, silu ? 1 : 0, M->A(o_off), s,
                                   hasraw ? M->A(r_off) : nullptr, fdt);
        }, (up ? 8.0 * 4 : 32.0 / 4) * Bn * H * Wd * C * outs,
           (double)dt_size(fdt) * Bn * H * Wd * C * (up ? 1.0 + 4.0 * outs : 1.0 + 0.25 * outs));
        return o;
    }

    // ResnetBlockBigGANpp.forward, layerspp.py:245-274
    Tn resblock(const Module& mod, const Tn& x1, const Tn* x2) {
        const float rs2 = 0.70710678118654752440f;
        Tn h1, xs;
        auto fusable_shape = [&](int dt, int H, int W, int C, int c2) {   // Conv(act(GroupNorm(t))) as one kernel for this shape?
            return dt != DT_F32 ? conv16_uses_halo(B, H, W, C, c2, mod.out_ch, 9)
                                : conv_supports_fused_gn(B, H, W, C, c2, mod.out_ch, 9);
        };
        auto fusable = [&](const Tn& t, int c2) { return fusable_shape(t.dt, t.H, t.W, t.C, c2); };
        // output geometry of the block (Conv_0 already runs at the resampled size)
        const int Ho = mod.up ? 2 * x1.H : mod.down ? x1.H / 2 : x1.H, Wo = mod.up ? 2 * x1.W : mod.down ? x1.W / 2 : x1.W;
        // Small images run split over K with a separate reduction launch.  Two of those launches disappear here:
        //  * Conv_0's reduction also finishes GroupNorm_1 (its group structure is known): it returns act(GN_1(.)) where
        //    the next conv wants a materialised input, or the pre-norm tensor plus per-channel mean / scale where the
        //    next conv normalises on load (GnFuse);
        //  * the shortcut Conv_2(x) leaves its slices to Conv_1's reduction, which sums both sets (SkPartial).
        GnFuse gf;
        gf.w_gamma = mod.w_gn1_g;
        gf.w_beta = mod.w_gn1_b;
        gf.silu = true;
        gf.apply = !fusable_shape(m->act_dt, Ho, Wo, mod.out_ch, 0);
        SkPartial sp;
        const bool merge_sc = mod.shortcut && sk_two_pass(x1.dt, Ho, Wo, mod.out_ch, mod.out_ch, 9) &&
                              sk_two_pass(x1.dt, Ho, Wo, mod.in_ch, mod.out_ch, 1);
        if (!mod.up && !mod.down) {
            if (mod.shortcut) {      // Conv_2(x) is independent of GN_0 -> Conv_0 -> GN_1: a side branch up to Conv_1's launch
                fork();
                xs = conv("conv2_1x1", x1, x2, mod.w_c2, mod.w_c2_b, -1, mod.out_ch, 1, nullptr, 1.f, false, false, nullptr,
                          false, -1, -1, merge_sc ? &sp : nullptr);
                side_end();
            }
            if (fusable(x1, x2 ? x2->C : 0)) {
                // Conv_0(act(GroupNorm_0(x))) in one kernel: the normalised tensor never reaches HBM
                GnBuf g0 = gn(x1, x2, mod.w_gn0_g, mod.w_gn0_b);
                h1 = conv("conv0_3x3_gn", x1, x2, mod.w_c0, -1, mod.dense_row0, mod.out_ch, 9, nullptr, 1.f, false,
                          false, &g0, true, mod.wq_c0, -1, nullptr, nullptr, &gf);
                gn_release(g0);
            } else {
                Tn h0 = gn_norm(x1, x2, mod.w_gn0_g, mod.w_gn0_b, true);
                h1 = conv("conv0_3x3", h0, nullptr, mod.w_c0, -1, mod.dense_row0, mod.out_ch, 9, nullptr, 1.f, false, false,
                          nullptr, false, -1, -1, nullptr, nullptr, &gf);
                release(h0);
            }
        } else {
            GnBuf g0 = gn(x1, x2, mod.w_gn0_g, mod.w_gn0_b);
            Tn xr;
            Tn hr = fir(x1, mod.up, &g0, true, nullptr, false, &xr);      // act(GN(x)) and x resampled in one pass
            gn_release(g0);
            // the shortcut Conv_2(x) (layerspp.py:268-270) does not depend on the GN -> Conv_0 -> GN chain: side branch
            fork();
            xs = conv("conv2_1x1", xr, nullptr, mod.w_c2, mod.w_c2_b, -1, mod.out_ch, 1, nullptr, 1.f, false, false, nullptr,
                      false, -1, -1, merge_sc ? &sp : nullptr);
            side_end();
            h1 = conv("conv0_3x3", hr, nullptr, mod.w_c0, -1, mod.dense_row0, mod.out_ch, 9, nullptr, 1.f, false, false,
                      nullptr, false, mod.wq_c0, -1, nullptr, nullptr, &gf);
            release(hr);
            release(xr);
        }
        Tn out;
        const bool forked = mod.shortcut;
        const Tn* resid = sp.valid ? nullptr : (xs.valid() ? &xs : &x1);
        const SkPartial* extra = sp.valid ? &sp : nullptr;
        if (gf.done && gf.apply) {                       // h1 already is act(GroupNorm_1(Conv_0(.)))
            if (forked) join();
            out = conv("conv1_3x3", h1, nullptr, mod.w_c1, mod.w_c1_b, -1, mod.out_ch, 9, resid, rs2, false, false, nullptr,
                       false, -1, -1, nullptr, extra);
            release(h1);
        } else if (fusable(h1, 0)) {
            GnBuf g1 = gf.done ? gf.g : gn(h1, nullptr, mod.w_gn1_g, mod.w_gn1_b);
            if (forked) join();
            out = conv("conv1_3x3_gn", h1, nullptr, mod.w_c1, mod.w_c1_b, -1, mod.out_ch, 9, resid, rs2,
                       false, false, &g1, true, mod.wq_c1, -1, nullptr, extra);
            gn_release(g1);
            release(h1);
        } else {
            Tn h2 = gn_norm(h1, nullptr, mod.w_gn1_g, mod.w_gn1_b, true);
            release(h1);
            if (forked) join();
            out = conv("conv1_3x3", h2, nullptr, mod.w_c1, mod.w_c1_b, -1, mod.out_ch, 9, resid, rs2, false, false, nullptr,
                       false, -1, -1, nullptr, extra);
            release(h2);
        }
        if (sp.valid) arena.release(sp.part_off);
        release(xs);
        return out;
    }
    // true when a conv of this shape runs split over K with the separate (two-pass) reduction launch
    bool sk_two_pass(int dt, int H, int W, int Cin, int Cout, int taps) const {
        if (conv_splitk_in_launch() || getenv("FLOWSE_NO_MERGED_REDUCE")) return false;
        const int ks = dt != DT_F32 ? ((conv_supports_head4(B, H, W, Cin, 0, Cout, taps) || conv16_uses_halo(B, H, W, Cin, 0, Cout, taps))
                                           ? 1 : conv16_ksplit(B, H, W, Cin, Cout, taps))
                                    : conv_ksplit(B, H, W, Cin, Cout, taps);
        return ks > 1;
    }

    // AttnBlockpp.forward, layerspp.py:75-91
    Tn attn(const Module& mod, const Tn& x) {
        flowse_model* M = m;
        const float rs2 = 0.70710678118654752440f;
        const int C = x.C, L = x.H * x.W, Bn = B;
        // the attention sub-block keeps fp32 intermediates in every mode (0.3 % of the FLOPs): GroupNorm widens, the
        // output projection rounds back to the activation type while adding the skip
        Tn hn = gn_norm(x, nullptr, mod.w_gn0_g, mod.w_gn0_b, false, DT_F32);
        Tn qkv = conv("attn_qkv", hn, nullptr, mod.w_qkv, mod.w_qkv_b, -1, 3 * C, 1, nullptr, 1.f, false, false, nullptr,
                      false, -1, DT_F32);
        release(hn);
        Tn o = alloc(x.H, x.W, C, DT_F32);
        const size_t q_off = qkv.off, o_off = o.off;
        op("attention@" + std::to_string(x.H) + "x" + std::to_string(x.W), [=](hipStream_t s) { return launch_attention(M->A(q_off), Bn, L, C, M->A(o_off), s); },
           4.0 * Bn * (double)L * L * C, 16.0 * Bn * L * C);
        release(qkv);
        Tn out = conv("attn_out", o, nullptr, mod.w_o, mod.w_o_b, -1, C, 1, &x, rs2, false, false, nullptr, false, -1,
                      x.dt);
        release(o);
        return out;
    }
};

// NCSNpp.forward, ncsnpp.py:247-404
static int build_plan(flowse_model* m, Plan* plan, int B, int F, int T) {
    const flowse_config& c = m->cfg;
    const int L = c.num_levels;
    if (F != c.image_size) {
        set_error("F=%d must equal image_size=%d (attention placement, ncsnpp.py:298)", F, c.image_size);
        return ERR_SHAPE;
    }
    if (B < 1 || T < 1 || (T % (1 << (L - 1))) != 0 || (F % (1 << (L - 1))) != 0) {
        set_error("shape B=%d F=%d T=%d: T and F must be multiples of %d (pad_spec)", B, F, T, 1 << (L - 1));
        return ERR_SHAPE;
    }
    plan->B = B; plan->F = F; plan->T = T;
    Builder bd;
    bd.m = m;
    bd.plan = plan;
    bd.B = B;
    bd.branches = m->use_branches;
    flowse_model* M = m;
    const int nf = c.nf, td = m->temb_dim;
    size_t mi = 0;
    auto next = [&]() -> const Module& { return m->mods[mi++]; };

    // ---- time embedding (depends only on t)
    const Module& gfp = next();
    const Module& lin1 = next();
    const Module& lin2 = next();
    const size_t e0 = bd.arena.alloc((size_t)B * 2 * nf * 4), e1 = bd.arena.alloc((size_t)B * td * 4),
                 e2 = bd.arena.alloc((size_t)B * td * 4);
    bd.M_table_off = bd.arena.alloc((size_t)B * m->dense_rows * 4);
    const size_t table = bd.M_table_off;
    {
        const int64_t wg = gfp.w_a, w1 = lin1.w_a, b1 = lin1.w_a_b, w2 = lin2.w_a, b2 = lin2.w_a_b;
        bd.op("gfp", [=](hipStream_t s) { return launch_gfp(nullptr, M->W(wg), B, nf, M->A(e0), s, M->d_call); });
        bd.op("temb_linear1", [=](hipStream_t s) {
            return launch_linear(M->A(e0), B, 2 * nf, M->W(w1), M->W(b1), td, 1, M->A(e1), td, s);
        });
        // act(temb) is the only consumer of temb (layerspp.py:263)
        bd.op("temb_linear2", [=](hipStream_t s) {
            return launch_linear(M->A(e1), B, td, M->W(w2), M->W(b2), td, 1, M->A(e2), td, s);
        });
        bd.op("dense_table", [=](hipStream_t s) {
            return launch_linear(M->A(e2), B, td, M->W(M->w_dense), M->W(M->w_dense_b), M->dense_rows, 0,
                                 M->A(table), M->dense_rows, s);
        });
    }
    // ---- feature pack + input conv
    Tn in4 = bd.alloc(F, T, 4);
    {
        const size_t o = in4.off;
        bd.op("pack_input", [=](hipStream_t s) { return launch_pack_input(nullptr, nullptr, B, F, T, M->A(o), s, M->d_call); });
    }
    const Module& cin = next();
    std::vector<Tn> hs;
    hs.push_back(bd.conv("conv_in", in4, nullptr, cin.w_a, cin.w_a_b, -1, nf, 9, nullptr, 1.f, false, true));
    Tn ipyr = in4;
    // ---- down path
    for (int lv = 0; lv < L; ++lv) {
        for (int b = 0; b < c.num_res_blocks; ++b) {
            Tn h = bd.resblock(next(), hs.back(), nullptr);
            if (in_list(c.attn_resolutions, c.num_attn, h.H)) {
                Tn h2 = bd.attn(next(), h);
                bd.release(h);
                h = h2;
            }
            hs.push_back(h);
        }
        if (lv != L - 1) {
            Tn h = bd.resblock(next(), hs.back(), nullptr);
            Tn ip2 = bd.fir(ipyr, false, nullptr, false, nullptr);
            bd.release(ipyr);
            ipyr = ip2;
            const Module& cb = next();
            bd.conv("combine_1x1", ipyr, nullptr, cb.w_a, cb.w_a_b, -1, cb.out_ch, 1, &h, 1.f, true, true);
            bd.drop_stats(h);          // h was updated in place
            hs.push_back(h);
        }
    }
    bd.release(ipyr);
    // ---- middle
    Tn h = bd.resblock(next(), hs.back(), nullptr);
    {
        Tn h2 = bd.attn(next(), h);
        bd.release(h);
        h = bd.resblock(next(), h2, nullptr);
        bd.release(h2);
    }
    // ---- up path
    Tn pyr;
    for (int lv = L - 1; lv >= 0; --lv) {
        for (int b = 0; b < c.num_res_blocks + 1; ++b) {
            Tn skip = hs.back();
            hs.pop_back();
            Tn h2 = bd.resblock(next(), h, &skip);
            bd.release(h);
            bd.release(skip);
            h = h2;
        }
        if (in_list(c.attn_resolutions, c.num_attn, h.H)) {
            Tn h2 = bd.attn(next(), h);
            bd.release(h);
            h = h2;
        }
        const Module& gnm = next();
        const Module& pcv = next();
        // 16-bit h: the dedicated 4-channel head kernel reads it directly; small images materialise act(GN(h)) in fp32
        // and run the fp32 kernels
        const bool pfuse = h.dt != DT_F32 ? conv_supports_head4(B, h.H, h.W, h.C, 0, 4, 9)
                                          : conv_supports_fused_gn(B, h.H, h.W, h.C, 0, 4, 9);
        GnBuf g;
        if (pfuse) g = bd.gn(h, nullptr, gnm.w_a, gnm.w_a_b);
        Tn ph = pfuse ? h : bd.gn_norm(h, nullptr, gnm.w_a, gnm.w_a_b, true, DT_F32);
        const GnBuf* pg = pfuse ? &g : nullptr;
        if (!pyr.valid()) {
            pyr = bd.conv("pyramid_conv", ph, nullptr, pcv.w_a, pcv.w_a_b, -1, 4, 9, nullptr, 1.f, false, false, pg, true);
        } else {
            Tn up = bd.fir(pyr, true, nullptr, false, nullptr);
            bd.release(pyr);
            pyr = bd.conv("pyramid_conv", ph, nullptr, pcv.w_a, pcv.w_a_b, -1, 4, 9, &up, 1.f, true, false, pg, true);
        }
        if (pfuse) bd.gn_release(g);
        if (!pfuse) bd.release(ph);
        if (lv != 0) {
            Tn h2 = bd.resblock(next(), h, nullptr);
            bd.release(h);
            h = h2;
        }
    }
    bd.release(h);
    if (!hs.empty() || mi != m->mods.size()) {
        set_error("internal: plan consumed %zu of %zu modules, %zu skips left", mi, m->mods.size(), hs.size());
        return ERR_STATE;
    }
    // ---- head (+ solver update)
    {
        const size_t p = pyr.off;
        bd.op("head", [=](hipStream_t s) {
            return launch_head(M->A(p), nullptr, M->W(M->w_out), M->W(M->w_out_b), B, F, T, 0, nullptr, 0.f, nullptr, s,
                               M->d_call);
        });
    }
    plan->ws_bytes = bd.arena.peak();
    return bd.failed ? ERR_STATE : OK;
}

static void drop_graph(Plan* p) {
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    p->exec = nullptr;
    p->graph = nullptr;
    p->eager_runs = 0;
}

static void clear_plans(flowse_model* m) {
    for (auto& kv : m->plans) drop_graph(&kv.second);
    m->plans.clear();
    m->block_plans.clear();
}

// every device call of a handle must be made with the handle's device current (the buffers live there)
static int check_device(const flowse_model* m) {
    int dev = 0;
    FLOWSE_HIP(hipGetDevice(&dev));
    if (m->device >= 0 && dev != m->device) {
        set_error("model handle is bound to HIP device %d but device %d is current (reload the weights on the new "
                  "device, or hipSetDevice back)", m->device, dev);
        return ERR_STATE;
    }
    return OK;
}

static int get_plan(flowse_model* m, int B, int F, int T, Plan** out) {
    if (!m->d_w) {
        set_error("weights not loaded: call flowse_model_load_weights first");
        return ERR_STATE;
    }
    if (const int rc = check_device(m)) return rc;
    auto key = std::make_tuple(B, F, T);
    auto it = m->plans.find(key);
    if (it == m->plans.end()) {
        Plan p;
        const int rc = build_plan(m, &p, B, F, T);
        if (rc != OK) return rc;
        it = m->plans.emplace(key, std::move(p)).first;
    }
    Plan* p = &it->second;
    if (p->ws_bytes > m->d_ws_bytes) {
        // growing the workspace: the stream may still be using the old one, and captured graphs point into it
        FLOWSE_HIP(hipDeviceSynchronize());
        for (auto& kv : m->plans) drop_graph(&kv.second);
        if (m->d_ws) FLOWSE_HIP(hipFree(m->d_ws));
        m->d_ws = nullptr;
        m->d_ws_bytes = 0;
        FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_ws), p->ws_bytes));
        m->d_ws_bytes = p->ws_bytes;
    }
    *out = p;
    return OK;
}

static int prof_event(flowse_model* m, hipEvent_t* e) {
    if (m->prof_used == m->prof_pool.size()) {
        hipEvent_t ev;
        FLOWSE_HIP(hipEventCreate(&ev));
        m->prof_pool.push_back(ev);
    }
    *e = m->prof_pool[m->prof_used++];
    return OK;
}

static int run_plan(flowse_model* m, Plan* p, hipStream_t s) {
    if (m->prof_mode != -1) {
        for (size_t i = 0; i < p->ops.size(); ++i) {
            m->prof_tot_flops += p->flops[i];
            m->prof_tot_issued += p->issued[i];
        }
        m->prof_tot_launches += (int64_t)p->ops.size();
    }
    // two-stream execution of the side branches (never while profiling -- per-launch events want one stream -- and never
    // on the NULL stream, which serialises against every blocking stream anyway)
    const bool multi = m->use_branches && m->prof_mode == -1 && s != nullptr;
    size_t ev_used = 0;
    auto next_event = [&](hipEvent_t* e) -> int {
        if (ev_used == m->br_events.size()) {
            hipEvent_t ev;
            FLOWSE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            m->br_events.push_back(ev);
        }
        *e = m->br_events[ev_used++];
        return OK;
    };
    if (multi && !m->sstream) FLOWSE_HIP(hipStreamCreateWithFlags(&m->sstream, hipStreamNonBlocking));
    bool side_open = false;                 // the side stream holds work the main stream has not joined yet
    for (size_t i = 0; i < p->ops.size(); ++i) {
        if (multi && p->sync[i]) {
            hipEvent_t ev;
            if (p->sync[i] & 2) {           // join: main waits for the side branch
                if (const int rc = next_event(&ev)) return rc;
                FLOWSE_HIP(hipEventRecord(ev, m->sstream));
                FLOWSE_HIP(hipStreamWaitEvent(s, ev, 0));
                side_open = false;
            }
            if (p->sync[i] & 1) {           // fork: the side branch starts after everything enqueued on main so far
                if (const int rc = next_event(&ev)) return rc;
                FLOWSE_HIP(hipEventRecord(ev, s));
                FLOWSE_HIP(hipStreamWaitEvent(m->sstream, ev, 0));
                side_open = true;
            }
        }
        const bool prof = m->prof_mode == 1 || (m->prof_mode == 0 && p->dominant[i]);
        flowse_model::Pending pd;
        if (prof) {
            const std::string& name = (m->prof_mode == 0) ? std::string("dominant_conv3x3") : p->labels[i];
            auto it = m->prof_label_ix.find(name);
            if (it == m->prof_label_ix.end()) {
                it = m->prof_label_ix.emplace(name, (int)m->prof_labels.size()).first;
                m->prof_labels.push_back(name);
            }
            pd.label = it->second;
            pd.flops = p->flops[i];
            pd.bytes = p->bytes[i];
            pd.issued = p->issued[i];
            int rc = prof_event(m, &pd.a);
            if (rc != OK) return rc;
            rc = prof_event(m, &pd.b);
            if (rc != OK) return rc;
            FLOWSE_HIP(hipEventRecord(pd.a, s));
        }
#ifdef FLOWSE_TS
        {   // measurement build only: FLOWSE_SKIP_OPS=prefix[,prefix...] drops launches by label (timing what-ifs)
            static const char* skip = getenv("FLOWSE_SKIP_OPS");
            bool drop = false;
            if (skip) {
                const std::string all(skip);
                size_t a = 0;
                while (a <= all.size()) {
                    size_t b = all.find(',', a);
                    if (b == std::string::npos) b = all.size();
                    if (b > a && p->labels[i].compare(0, b - a, all, a, b - a) == 0) drop = true;
                    a = b + 1;
                }
            }
            if (drop) continue;
        }
#endif
        const int rc = p->ops[i]((multi && p->side[i]) ? m->sstream : s);
        if (rc != OK) return rc;
        if (prof) {
            FLOWSE_HIP(hipEventRecord(pd.b, s));
            m->prof_pending.push_back(pd);
        }
    }
    if (multi && side_open) {               // (every fork is joined by construction; belt and braces for stream capture)
        hipEvent_t ev;
        if (const int rc = next_event(&ev)) return rc;
        FLOWSE_HIP(hipEventRecord(ev, m->sstream));
        FLOWSE_HIP(hipStreamWaitEvent(s, ev, 0));
    }
    return OK;
}

// One network evaluation.  First call per shape: eager (also performs the one-time per-device kernel attribute
// setup); second call: the same launch list is captured into a hipGraph; afterwards one hipGraphLaunch per call.
// Profiling (per-launch events), the NULL stream and FLOWSE_NO_GRAPH=1 keep the eager path.
static int exec_plan(flowse_model* m, Plan* p, hipStream_t s) {
    if (!m->use_graph || m->prof_mode != -1 || s == nullptr) return run_plan(m, p, s);   // (NULL: see enter_stream)
    if (p->exec) {
        FLOWSE_HIP(hipGraphLaunch(p->exec, s));
        ++m->graph_launches;
        return OK;
    }
    if (p->eager_runs < 1) {
        ++p->eager_runs;
        return run_plan(m, p, s);
    }
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();                    // e.g. the caller's stream is already capturing: stay eager
        m->use_graph = false;
        return run_plan(m, p, s);
    }
    const int rc = run_plan(m, p, s);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) {                    // capture refused: stay eager for this plan
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        m->use_graph = false;
        return run_plan(m, p, s);
    }
    hipGraphExec_t ex = nullptr;
    if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess || !ex) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(g);
        m->use_graph = false;
        return run_plan(m, p, s);
    }
    p->graph = g;
    p->exec = ex;
    FLOWSE_HIP(hipGraphLaunch(p->exec, s));
    ++m->graph_launches;
    return OK;
}

// Stream the work of one C-ABI call runs on.  A real stream: that stream.  The NULL stream: it cannot be captured, so
// (unless graphs are off / a profile is being taken) the call moves to the handle's internal stream, which first waits
// for everything the caller has enqueued on the NULL stream; leave_stream() makes the NULL stream wait for the call.
static int enter_stream(flowse_model* m, hipStream_t caller, hipStream_t* work) {
    *work = caller;
    if (caller != nullptr || !(m->use_graph || m->use_branches) || m->prof_mode != -1) return OK;
    if (!m->gstream) {
        FLOWSE_HIP(hipStreamCreateWithFlags(&m->gstream, hipStreamNonBlocking));
        FLOWSE_HIP(hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming));
        FLOWSE_HIP(hipEventCreateWithFlags(&m->ev_out, hipEventDisableTiming));
    }
    FLOWSE_HIP(hipEventRecord(m->ev_in, nullptr));
    FLOWSE_HIP(hipStreamWaitEvent(m->gstream, m->ev_in, 0));
    *work = m->gstream;
    return OK;
}
static int leave_stream(flowse_model* m, hipStream_t caller, hipStream_t work) {
    if (work == caller) return OK;
    FLOWSE_HIP(hipEventRecord(m->ev_out, work));
    FLOWSE_HIP(hipStreamWaitEvent(caller, m->ev_out, 0));
    return OK;
}

// Plan of a single-module handle: inputs are copied into the arena, the module runs exactly as inside the network
// (Builder::resblock / attn / conv), the result is copied out.
static int build_block_plan(flowse_model* m, Plan* plan, int B, int H, int W, int C1) {
    const Module& mod = m->mods[0];
    const bool combine = mod.kind == M_COMBINE;
    const int C2 = combine ? mod.out_ch : mod.in_ch - C1;
    if (B < 1 || H < 1 || W < 1 || C1 < 4 || (C1 & 3) || C2 < 0 || (C2 & 3) || (combine && C1 != 4) ||
        (mod.kind == M_ATTN && C2 != 0) || ((mod.up || mod.down) && C2 != 0) || (mod.down && ((H | W) & 1))) {
        set_error("flowse_block_forward: bad shape B=%d H=%d W=%d C1=%d for a module with in_ch=%d", B, H, W, C1,
                  mod.in_ch);
        return ERR_SHAPE;
    }
    plan->B = B; plan->F = H; plan->T = W;
    Builder bd;
    bd.m = m;
    bd.plan = plan;
    bd.B = B;
    bd.branches = m->use_branches;
    flowse_model* M = m;
    const int td = m->temb_dim;
    Tn x1 = bd.alloc(H, W, C1), x2;
    if (C2 > 0) x2 = bd.alloc(H, W, C2);
    {   // the caller's tensors are fp32; in a 16-bit storage mode they are rounded to the activation type on the way in
        const size_t o1 = x1.off, o2 = x2.off;
        const int64_t n1 = (int64_t)B * H * W * C1, n2 = C2 > 0 ? (int64_t)B * H * W * C2 : 0;
        const int d1 = x1.dt, d2 = x2.dt;
        bd.op("block_in", [=](hipStream_t s) {
            int rc = launch_convert(M->bcall.in1, DT_F32, M->A(o1), d1, n1, s);
            if (rc == OK && n2) rc = launch_convert(M->bcall.in2, DT_F32, M->A(o2), d2, n2, s);
            return rc;
        });
    }
    Tn out;
    if (mod.kind == M_RESBLOCK) {
        bd.M_table_off = bd.arena.alloc((size_t)B * m->dense_rows * 4);
        const size_t table = bd.M_table_off;
        bd.op("dense_table", [=](hipStream_t s) {          // Dense_0(act(temb)) + Conv_0.bias (layerspp.py:262-263)
            return launch_linear(M->bcall.temb_act, B, td, M->W(M->w_dense), M->W(M->w_dense_b), M->dense_rows, 0,
                                 M->A(table), M->dense_rows, s);
        });
        out = bd.resblock(mod, x1, C2 > 0 ? &x2 : nullptr);
    } else if (mod.kind == M_ATTN) {
        out = bd.attn(mod, x1);
    } else {                                               // Combine: conv1x1(x) + y (layerspp.py:55-59)
        bd.conv("combine_1x1", x1, nullptr, mod.w_a, mod.w_a_b, -1, mod.out_ch, 1, &x2, 1.f, true, true);
        out = x2;
    }
    {
        const size_t o = out.off;
        const int64_t n = (int64_t)out.B * out.H * out.W * out.C;
        const int od = out.dt;
        bd.op("block_out", [=](hipStream_t s) { return launch_convert(M->A(o), od, M->bcall.out, DT_F32, n, s); });
    }
    plan->ws_bytes = bd.arena.peak();
    return bd.failed ? ERR_STATE : OK;
}

static void free_device_state(flowse_model* m) {
    int cur = 0;
    const bool sw = m->device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != m->device;
    if (sw) (void)hipSetDevice(m->device);
    if (m->device >= 0) (void)hipDeviceSynchronize();
    clear_plans(m);
    if (m->d_w) (void)hipFree(m->d_w);
    if (m->d_ws) (void)hipFree(m->d_ws);
    if (m->d_ts) (void)hipFree(m->d_ts);
    if (m->d_wq) (void)hipFree(m->d_wq);
    if (m->d_w16) (void)hipFree(m->d_w16);
    if (m->d_wino) (void)hipFree(m->d_wino);
    if (m->d_call) (void)hipFree(m->d_call);
    if (m->d_ticket) (void)hipFree(m->d_ticket);
    m->d_ticket = nullptr;
    if (m->d_rk) (void)hipFree(m->d_rk);
    m->d_rk = nullptr;
    m->d_rk_floats = 0;
    if (m->sstream) (void)hipStreamDestroy(m->sstream);
    m->sstream = nullptr;
    for (hipEvent_t e : m->br_events) (void)hipEventDestroy(e);
    m->br_events.clear();
    if (m->gstream) (void)hipStreamDestroy(m->gstream);
    if (m->ev_in) (void)hipEventDestroy(m->ev_in);
    if (m->ev_out) (void)hipEventDestroy(m->ev_out);
    m->gstream = nullptr;
    m->ev_in = m->ev_out = nullptr;
    for (hipEvent_t e : m->prof_pool) (void)hipEventDestroy(e);
    m->prof_pool.clear();
    m->prof_used = 0;
    m->d_w = nullptr; m->d_ws = nullptr; m->d_ts = nullptr; m->d_wino = nullptr; m->d_call = nullptr;
    m->d_wq = nullptr;
    m->d_w16 = nullptr;
    m->d_w_numel = m->d_wq_numel = m->d_wino_numel = m->d_w16_numel = 0;
    m->d_ws_bytes = m->d_ts_floats = 0;
    m->device = -1;
    if (sw) (void)hipSetDevice(cur);
}

}  // namespace flowse

// =============================================================================================== C ABI
extern "C" {

int flowse_abi_version(void) { return FLOWSE_ABI_VERSION; }
const char* flowse_last_error(void) { return g_err; }

int flowse_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int flowse_model_create(const flowse_config* cfg, flowse_model** out) {
    if (!cfg || !out) {
        set_error("flowse_model_create: null argument");
        return ERR_ARG;
    }
    flowse_model* m = new flowse_model();
    m->cfg = *cfg;
    // hipGraph replay of the launch list is opt-in (FLOWSE_GRAPH=1): measured on MI355X / ROCm 7.2 a replayed graph of
    // ~400 short kernel nodes runs 5 % SLOWER than the same launches issued eagerly from the C loop at [1,1,256,256]
    // (8.09 k vs 8.55 k frames/s) and equal at [8,1,256,256]; the host is nowhere near launch-bound (~1.5 ms of launch
    // calls per 6 ms network evaluation at batch 1).  FLOWSE_NO_GRAPH=1 is accepted for compatibility.
    m->use_graph = getenv("FLOWSE_GRAPH") != nullptr && getenv("FLOWSE_NO_GRAPH") == nullptr;
    // Two-stream execution of the shortcut branch is opt-in as well (FLOWSE_BRANCH=1): the fork / join events cost more
    // than the overlap returns -- 17.05 k vs 17.5 k frames/s at [8,1,256,256], 8.2 k vs 8.65 k at [1,1,256,256].
    m->use_branches = getenv("FLOWSE_BRANCH") != nullptr;
    const int rc = build_structure(m);
    if (rc != OK) {
        delete m;
        return rc;
    }
    *out = m;
    return OK;
}

void flowse_model_destroy(flowse_model* m) {
    if (!m) return;
    free_device_state(m);
    delete m;
}

int flowse_block_create(int kind, int in_ch, int out_ch, int up, int down, int temb_dim, flowse_model** out) {
    if (!out || kind < FLOWSE_BLOCK_RESNET || kind > FLOWSE_BLOCK_COMBINE || in_ch < 4 || (in_ch & 3) || out_ch < 4 ||
        (out_ch & 3) || (up && down) || (kind == FLOWSE_BLOCK_RESNET && temb_dim < 1) ||
        (kind == FLOWSE_BLOCK_ATTN && in_ch != out_ch) || (kind == FLOWSE_BLOCK_COMBINE && in_ch != 4)) {
        set_error("flowse_block_create: bad argument (kind=%d in_ch=%d out_ch=%d up=%d down=%d temb_dim=%d)", kind, in_ch,
                  out_ch, up, down, temb_dim);
        return ERR_ARG;
    }
    flowse_model* m = new flowse_model();
    memset(&m->cfg, 0, sizeof(m->cfg));
    m->use_graph = false;
    m->use_branches = getenv("FLOWSE_BRANCH") != nullptr;
    m->block_kind = kind;
    m->temb_dim = temb_dim;
    if (kind == FLOWSE_BLOCK_RESNET) add_module(m, resblock(in_ch, out_ch, up != 0, down != 0));
    else add_module(m, simple(kind == FLOWSE_BLOCK_ATTN ? M_ATTN : M_COMBINE, in_ch, out_ch));
    *out = m;
    return OK;
}

int flowse_block_forward(flowse_model* m, const float* in1, int C1, const float* in2, const float* temb_act, float* out,
                         int B, int H, int W, void* stream) {
    if (!m || m->block_kind < 0 || !in1 || !out || (m->block_kind == FLOWSE_BLOCK_RESNET && !temb_act) ||
        (m->block_kind == FLOWSE_BLOCK_COMBINE && !in2)) {
        set_error("flowse_block_forward: bad argument (not a block handle, or a required pointer is null)");
        return ERR_ARG;
    }
    if (!m->d_w) {
        set_error("weights not loaded: call flowse_model_load_weights first");
        return ERR_STATE;
    }
    if (const int rc = check_device(m)) return rc;
    if (!in2 && m->block_kind == FLOWSE_BLOCK_RESNET) C1 = m->mods[0].in_ch;
    auto key = std::make_tuple(B, H, W, C1);
    auto it = m->block_plans.find(key);
    if (it == m->block_plans.end()) {
        Plan p;
        const int rc = build_block_plan(m, &p, B, H, W, C1);
        if (rc != OK) return rc;
        it = m->block_plans.emplace(key, std::move(p)).first;
    }
    Plan* p = &it->second;
    if (p->ws_bytes > m->d_ws_bytes) {
        FLOWSE_HIP(hipDeviceSynchronize());
        if (m->d_ws) FLOWSE_HIP(hipFree(m->d_ws));
        m->d_ws = nullptr;
        m->d_ws_bytes = 0;
        FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_ws), p->ws_bytes));
        m->d_ws_bytes = p->ws_bytes;
    }
    m->bcall.in1 = in1;
    m->bcall.in2 = in2;
    m->bcall.temb_act = temb_act;
    m->bcall.out = out;
    return run_plan(m, p, static_cast<hipStream_t>(stream));
}

int flowse_model_num_params(const flowse_model* m) { return m ? (int)m->params.size() : 0; }
int flowse_model_num_modules(const flowse_model* m) { return m ? (int)m->mods.size() : 0; }
int64_t flowse_model_blob_numel(const flowse_model* m) { return m ? m->blob_numel : 0; }

int flowse_model_param_info(const flowse_model* m, int index, char* name, int name_cap, int64_t shape[4], int* ndim,
                            int64_t* offset) {
    if (!m || index < 0 || index >= (int)m->params.size()) {
        set_error("flowse_model_param_info: bad index %d", index);
        return ERR_ARG;
    }
    const ParamInfo& p = m->params[index];
    if (name && name_cap > 0) {
        strncpy(name, p.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (shape)
        for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
    if (ndim) *ndim = p.ndim;
    if (offset) *offset = p.offset;
    return OK;
}

int flowse_model_set_precision(flowse_model* m, int mode) {
    if (!m || mode < 0 || mode > 3) {
        set_error("flowse_model_set_precision: mode must be 0 (fp32), 1 (bf16x3), 2 (bf16) or 3 (fp16)");
        return ERR_ARG;
    }
    if (mode != m->precision) {
        if (m->d_w) {            // weights must be re-uploaded so that the operand planes match the mode
            if (const int rc = check_device(m)) return rc;      // before any state changes: a failure leaves the handle as is
            FLOWSE_HIP(hipDeviceSynchronize());
            clear_plans(m);
            FLOWSE_HIP(hipFree(m->d_w));
            m->d_w = nullptr;
            m->d_w_numel = 0;
            if (m->d_w16) {      // the 16-bit twin belongs to the mode that is being left
                FLOWSE_HIP(hipFree(m->d_w16));
                m->d_w16 = nullptr;
                m->d_w16_numel = 0;
            }
        }
        m->precision = mode;
        m->act_dt = DT_F32;      // recomputed by the next flowse_model_load_weights
        clear_plans(m);
    }
    return OK;
}

int flowse_model_load_weights(flowse_model* m, const float* blob, int64_t numel) {
    if (!m || !blob) {
        set_error("flowse_model_load_weights: null argument");
        return ERR_ARG;
    }
    if (numel != m->blob_numel) {
        set_error("flowse_model_load_weights: blob has %lld floats, model needs %lld", (long long)numel,
                  (long long)m->blob_numel);
        return ERR_ARG;
    }
    Packer pk;
    const int rc = pack_weights(m, blob, pk);
    if (rc != OK) return rc;
    int dev = 0;
    FLOWSE_HIP(hipGetDevice(&dev));
    if (m->device >= 0 && m->device != dev) free_device_state(m);      // the handle moves to the current device
    m->device = dev;
    FLOWSE_HIP(hipDeviceSynchronize());
    clear_plans(m);          // closures captured weight offsets of the previous packing
    if (!m->d_call) FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_call), sizeof(CallBlock)));
    if (!m->d_ticket) {
        FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_ticket), SK_TICKETS * sizeof(unsigned)));
        FLOWSE_HIP(hipMemset(m->d_ticket, 0, SK_TICKETS * sizeof(unsigned)));
    }
    if (m->d_w && m->d_w_numel < (int64_t)pk.host.size()) {
        FLOWSE_HIP(hipFree(m->d_w));
        m->d_w = nullptr;
    }
    if (!m->d_w) {
        FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_w), pk.host.size() * sizeof(float)));
        m->d_w_numel = (int64_t)pk.host.size();
    }
    FLOWSE_HIP(hipMemcpy(m->d_w, pk.host.data(), pk.host.size() * sizeof(float), hipMemcpyHostToDevice));
    m->act_dt = storage_type_for(m);
    if (m->storage16()) {                    // elementwise 16-bit twin of the packed blob (conv weights keep their offsets)
        const int64_t n16 = ((int64_t)pk.host.size() + 3) & ~(int64_t)3;
        if (m->d_w16 && m->d_w16_numel < n16) {
            FLOWSE_HIP(hipFree(m->d_w16));
            m->d_w16 = nullptr;
        }
        if (!m->d_w16) {
            FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_w16), n16 * sizeof(uint16_t)));
            m->d_w16_numel = n16;
        }
        const int crc = launch_convert(m->d_w, DT_F32, m->d_w16, m->act_dt, (int64_t)pk.host.size() & ~(int64_t)3, nullptr);
        if (crc != OK) return crc;
        pk.wino.clear();                     // no fp32 Winograd kernels run on 16-bit activations
    }
    // F(2,3) Winograd weights, derived on the device from the packed fp32 weights just uploaded
    m->wino_of.clear();
    int64_t wino_total = 0;
    for (auto& r : pk.wino) {
        m->wino_of[r.off] = wino_total;
        wino_total += (conv_wino_numel(r.Cout, r.Cin) + 63) & ~(int64_t)63;
    }
    if (m->d_wino && m->d_wino_numel < wino_total) {
        FLOWSE_HIP(hipFree(m->d_wino));
        m->d_wino = nullptr;
    }
    if (!m->d_wino && wino_total > 0) {
        FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_wino), wino_total * sizeof(float)));
        m->d_wino_numel = wino_total;
    }
    for (auto& r : pk.wino) {
        float* dst = m->d_wino + m->wino_of[r.off];
        const int wrc = conv_wino_default_f43() ? launch_f43_weights(m->d_w + r.off, r.Cout, r.Cin, dst, nullptr)
                                                : launch_wino_weights(m->d_w + r.off, r.Cout, r.Cin, dst, nullptr);
        if (wrc != OK) return wrc;
    }
    FLOWSE_HIP(hipDeviceSynchronize());
    // optional bf16 planes for the 3x3 ResBlock convolutions the halo kernel can take
    for (auto& mod : m->mods) mod.wq_c0 = mod.wq_c1 = -1;
    if (m->precision != 0 && !m->storage16()) {
        const int terms = m->precision == 1 ? 3 : 1;
        std::vector<uint16_t> q;
        for (auto& mod : m->mods) {
            if (mod.kind != M_RESBLOCK || (mod.out_ch % 128) != 0) continue;
            if ((mod.in_ch % 32) == 0) {
                mod.wq_c0 = (int64_t)q.size();
                q.resize(q.size() + conv_bf16_numel(mod.out_ch, mod.in_ch, terms));
                pack_conv_bf16(blob + m->params[mod.p0 + 2].offset, mod.out_ch, mod.in_ch, terms, q.data() + mod.wq_c0,
                               m->precision == 3);
            }
            mod.wq_c1 = (int64_t)q.size();
            q.resize(q.size() + conv_bf16_numel(mod.out_ch, mod.out_ch, terms));
            pack_conv_bf16(blob + m->params[mod.p0 + 8].offset, mod.out_ch, mod.out_ch, terms, q.data() + mod.wq_c1,
                           m->precision == 3);
        }
        if (m->d_wq && m->d_wq_numel < (int64_t)q.size()) {
            FLOWSE_HIP(hipFree(m->d_wq));
            m->d_wq = nullptr;
        }
        if (!m->d_wq && !q.empty()) {
            FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_wq), q.size() * sizeof(uint16_t)));
            m->d_wq_numel = (int64_t)q.size();
        }
        if (!q.empty())
            FLOWSE_HIP(hipMemcpy(m->d_wq, q.data(), q.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    return OK;
}

int flowse_model_reserve(flowse_model* m, int B, int F, int T, int64_t* workspace_bytes) {
    if (!m) {
        set_error("flowse_model_reserve: null model");
        return ERR_ARG;
    }
    Plan* p = nullptr;
    const int rc = get_plan(m, B, F, T, &p);
    if (rc != OK) return rc;
    if (workspace_bytes) *workspace_bytes = (int64_t)p->ws_bytes;
    return OK;
}

int flowse_vf_forward(flowse_model* m, const void* x, const void* y, const float* t, void* out, int B, int F, int T,
                      int mode, void* stream) {
    if (!m || !x || !y || !t || !out || (mode != 0 && mode != 1)) {
        set_error("flowse_vf_forward: bad argument");
        return ERR_ARG;
    }
    Plan* p = nullptr;
    int rc = get_plan(m, B, F, T, &p);
    if (rc != OK) return rc;
    hipStream_t caller = static_cast<hipStream_t>(stream), s = nullptr;
    rc = enter_stream(m, caller, &s);
    if (rc != OK) return rc;
    CallBlock cb{static_cast<const float*>(x), static_cast<const float*>(y), t, static_cast<float*>(out), mode, 0.f};
    rc = launch_set_call(m->d_call, cb, s);
    if (rc == OK) rc = exec_plan(m, p, s);
    const int rc2 = leave_stream(m, caller, s);
    return rc != OK ? rc : rc2;
}

int flowse_prior_sample(const void* y, const void* z, float sigma, void* x_out, int64_t numel_complex, void* stream) {
    if (!y || !z || !x_out || numel_complex < 0) {
        set_error("flowse_prior_sample: bad argument");
        return ERR_ARG;
    }
    return launch_axpy(static_cast<const float*>(y), static_cast<const float*>(z), sigma, 2 * numel_complex,
                       static_cast<float*>(x_out), static_cast<hipStream_t>(stream));
}

int flowse_axpy(const void* x, const void* k, float dt, void* out, int64_t numel_complex, void* stream) {
    return flowse_prior_sample(x, k, dt, out, numel_complex, stream);
}

static int reserve_times(flowse_model* m, size_t need) {
    if (need > m->d_ts_floats) {
        FLOWSE_HIP(hipDeviceSynchronize());
        if (m->d_ts) FLOWSE_HIP(hipFree(m->d_ts));
        m->d_ts = nullptr;
        m->d_ts_floats = 0;
        FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_ts), need * sizeof(float)));
        m->d_ts_floats = need;
    }
    return OK;
}

int flowse_euler_sample(flowse_model* m, void* x_inout, const void* y, const float* ts, const float* dts, int N, int B,
                        int F, int T, void* stream) {
    return flowse_rk_sample(m, x_inout, y, ts, dts, N, FLOWSE_TABLEAU_EULER, B, F, T, stream);
}

// Fixed-step explicit Runge-Kutta over the reference's grid (see include/flowse_hip.h).  With v = dnn(cat[x, y], t)
// (so VF = -v) and h = dts[i] > 0 a step from t to t - h is
//   euler:  x += h v(x, t)
//   heun:   v1 = v(x, t), v2 = v(x + h v1, t - h);                    x += h/2 (v1 + v2)
//   rk4:    v1 = v(x, t), v2 = v(x + h/2 v1, t - h/2), v3 = v(x + h/2 v2, t - h/2), v4 = v(x + h v3, t - h);
//           x += h/6 (v1 + 2 v2 + 2 v3 + v4)
// Every stage is one network evaluation whose head kernel (mode 3) writes the next stage's input and folds the slope
// into the accumulator; no separate axpy launches, no host synchronisation.
int flowse_rk_sample(flowse_model* m, void* x_inout, const void* y, const float* ts, const float* dts, int N, int tableau,
                     int B, int F, int T, void* stream) {
    if (!m || !x_inout || !y || !ts || !dts || N < 1 || tableau < FLOWSE_TABLEAU_EULER || tableau > FLOWSE_TABLEAU_RK4) {
        set_error("flowse_rk_sample / flowse_euler_sample: bad argument");
        return ERR_ARG;
    }
    Plan* p = nullptr;
    int rc = get_plan(m, B, F, T, &p);
    if (rc != OK) return rc;
    const int stages = tableau == FLOWSE_TABLEAU_RK4 ? 4 : tableau == FLOWSE_TABLEAU_HEUN ? 2 : 1;
    // a step that ends at (or numerically below) t = 0 is the reference's own Euler update: the field divides by t and
    // embeds log t, so no stage may be evaluated at the end point of such a step (it is the LAST step of the reference's
    // grid, whose length equals the last grid time, sampling/__init__.py:53)
    std::vector<float> nfe_t;
    std::vector<int> step_stages(N);
    for (int i = 0; i < N; ++i) {
        const float t = ts[i], h = dts[i];
        const bool lands = (double)t - (double)h <= 1e-6 * std::max(1.0, std::fabs((double)t));
        const int st = lands ? 1 : stages;
        step_stages[i] = st;
        const float dt = -h;
        nfe_t.push_back(t);
        if (st == 2) nfe_t.push_back(t + dt);
        if (st == 4) {
            const float th = t + 0.5f * dt;
            nfe_t.push_back(th);
            nfe_t.push_back(th);
            nfe_t.push_back(t + dt);
        }
    }
    rc = reserve_times(m, nfe_t.size() * (size_t)B);
    if (rc != OK) return rc;
    const size_t state = (size_t)2 * B * F * T;                 // floats of one complex64 [B,1,F,T] tensor
    if (stages > 1 && m->d_rk_floats < 2 * state) {
        FLOWSE_HIP(hipDeviceSynchronize());
        if (m->d_rk) FLOWSE_HIP(hipFree(m->d_rk));
        m->d_rk = nullptr;
        m->d_rk_floats = 0;
        FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_rk), 2 * state * sizeof(float)));
        m->d_rk_floats = 2 * state;
    }
    hipStream_t caller = static_cast<hipStream_t>(stream), s = nullptr;
    rc = enter_stream(m, caller, &s);
    if (rc != OK) return rc;
    // vec_t = ones(B) * t (sampling/__init__.py:55), written on the device by a kernel that receives the times by value
    rc = launch_fill_times(m->d_ts, nfe_t.data(), (int)nfe_t.size(), B, s);
    float* const x = static_cast<float*>(x_inout);
    const float* const yy = static_cast<const float*>(y);
    float* const xs = m->d_rk;                                   // stage input
    float* const acc = m->d_rk ? m->d_rk + state : nullptr;      // x + sum_j b_j h v_j so far
    size_t k = 0;                                                // index of the next network evaluation
    auto stage = [&](const float* in, float* out, const float* acc_in, float* acc_out, float a, float b) {
        CallBlock cb{in, yy, m->d_ts + (k++) * B, out, 3, 0.f, x, acc_in, acc_out, a, b};
        int r = launch_set_call(m->d_call, cb, s);
        if (r == OK) r = exec_plan(m, p, s);
        return r;
    };
    for (int i = 0; i < N && rc == OK; ++i) {
        const float h = dts[i];
        if (step_stages[i] == 1) {
            CallBlock cb{x, yy, m->d_ts + (k++) * B, x, 2, h};
            rc = launch_set_call(m->d_call, cb, s);
            if (rc == OK) rc = exec_plan(m, p, s);
        } else if (step_stages[i] == 2) {
            rc = stage(x, xs, x, acc, h, 0.5f * h);
            if (rc == OK) rc = stage(xs, nullptr, acc, x, 0.f, 0.5f * h);
        } else {
            rc = stage(x, xs, x, acc, 0.5f * h, h / 6.0f);
            if (rc == OK) rc = stage(xs, xs, acc, acc, 0.5f * h, h / 3.0f);
            if (rc == OK) rc = stage(xs, xs, acc, acc, h, h / 3.0f);
            if (rc == OK) rc = stage(xs, nullptr, acc, x, 0.f, h / 6.0f);
        }
    }
    const int rc2 = leave_stream(m, caller, s);
    return rc != OK ? rc : rc2;
}

int64_t flowse_model_graph_launches(const flowse_model* m) { return m ? m->graph_launches : 0; }

int flowse_stft_compress(const float* sig, int B, int L, float scale_in, void* out_c64, int T, int Tpad, float factor,
                         float exponent, void* stream) {
    if (!sig || !out_c64) {
        set_error("flowse_stft_compress: null argument");
        return ERR_ARG;
    }
    return launch_stft_compress(sig, B, L, scale_in, static_cast<float*>(out_c64), T, Tpad, factor, exponent,
                                static_cast<hipStream_t>(stream));
}

int flowse_istft_decompress(const void* spec_c64, int B, int T, int Tpad, float factor, float exponent, float* out,
                            int Lout, float scale_out, void* stream) {
    if (!spec_c64 || !out) {
        set_error("flowse_istft_decompress: null argument");
        return ERR_ARG;
    }
    return launch_istft_decompress(static_cast<const float*>(spec_c64), B, T, Tpad, factor, exponent, out, Lout,
                                   scale_out, static_cast<hipStream_t>(stream));
}

int flowse_profile_begin(flowse_model* m, int mode) {
    if (!m || (mode != 0 && mode != 1)) {
        set_error("flowse_profile_begin: bad argument");
        return ERR_ARG;
    }
    m->prof_mode = mode;
    m->prof_used = 0;
    m->prof_pending.clear();
    m->prof_labels.clear();
    m->prof_label_ix.clear();
    m->prof_tot_flops = m->prof_tot_issued = 0.0;
    m->prof_tot_launches = 0;
    return OK;
}

int flowse_profile_end(flowse_model* m, char* json, int cap) {
    if (!m || !json || cap < 64) {
        set_error("flowse_profile_end: bad argument");
        return ERR_ARG;
    }
    m->prof_mode = -1;
    std::vector<ProfAcc> acc(m->prof_labels.size());
    for (auto& pd : m->prof_pending) {
        FLOWSE_HIP(hipEventSynchronize(pd.b));
        float ms = 0.f;
        FLOWSE_HIP(hipEventElapsedTime(&ms, pd.a, pd.b));
        ProfAcc& a = acc[pd.label];
        a.launches += 1;
        a.ms += ms;
        a.flops += pd.flops;
        a.bytes += pd.bytes;
        a.issued += pd.issued;
    }
    m->prof_pending.clear();
    m->prof_used = 0;
    std::string out = "{";
    for (size_t i = 0; i < acc.size(); ++i) {
        char buf[384];
        snprintf(buf, sizeof(buf),
                 "%s\"%s\": {\"launches\": %lld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e, \"issued\": %.6e}",
                 i ? ", " : "", m->prof_labels[i].c_str(), (long long)acc[i].launches, acc[i].ms, acc[i].flops,
                 acc[i].bytes, acc[i].issued);
        out += buf;
    }
    {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s\"_all_launches\": {\"launches\": %lld, \"ms\": 0, \"flops\": %.6e, \"bytes\": 0, "
                 "\"issued\": %.6e}", acc.empty() ? "" : ", ", (long long)m->prof_tot_launches, m->prof_tot_flops,
                 m->prof_tot_issued);
        out += buf;
    }
    out += "}";
    if ((int)out.size() + 1 > cap) {
        set_error("flowse_profile_end: report needs %zu bytes", out.size() + 1);
        return ERR_ARG;
    }
    memcpy(json, out.c_str(), out.size() + 1);
    return OK;
}

int flowse_upfirdn2d(const float* input, const float* kernel, int planes, int in_h, int in_w, int kh, int kw, int up_x,
                     int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float* out,
                     int out_h, int out_w, void* stream) {
    if (!input || !kernel || !out || pad_x0 < 0 || pad_x1 < 0 || pad_y0 < 0 || pad_y1 < 0) {
        set_error("flowse_upfirdn2d: bad argument (null pointer or negative pad)");
        return ERR_ARG;
    }
    return launch_upfirdn2d_nchw(input, kernel, planes, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                 pad_y0, pad_y1, out, out_h, out_w, static_cast<hipStream_t>(stream));
}

int64_t flowse_op_conv2d_scratch_floats(int B, int H, int W, int Cin, int Cout, int taps) {
    const int ks = conv_ksplit(B, H, W, Cin, Cout, taps);
    return ks > 1 ? (int64_t)ks * B * H * W * Cout : 0;
}

int flowse_op_conv2d(const float* in1, int C1, const float* in2, int C2, const float* w, const float* bias,
                     const float* bias2, int bias2_stride, const float* res, float* out, int B, int H, int W, int Cout,
                     int taps, float scale, float* splitk_scratch, void* stream) {
    if (!in1 || !w || !out) {
        set_error("flowse_op_conv2d: null argument");
        return ERR_ARG;
    }
    ConvArgs c;
    c.in1 = in1; c.in2 = in2; c.C1 = C1; c.C2 = in2 ? C2 : 0;
    c.w = w; c.bias = bias; c.bias2 = bias2; c.bias2_stride = bias2_stride; c.res = res; c.out = out;
    c.B = B; c.H = H; c.W = W; c.Cout = Cout; c.taps = taps; c.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C1 == 4 && !in2) return launch_conv_cin4(c, s);
    if (splitk_scratch) {
        c.ksplit = conv_ksplit(B, H, W, c.C1 + c.C2, Cout, taps);
        c.partial = c.ksplit > 1 ? splitk_scratch : nullptr;
    }
    return launch_conv(c, s);
}

int64_t flowse_op_group_norm_scratch_floats(int B, int HW, int C) {
    const int nblk = gn_partial_blocks(HW, C);
    return (int64_t)B * nblk * C * 2 + 2 * (int64_t)B * C;
}

int flowse_op_group_norm(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                         float eps, int silu, float* out, int B, int H, int W, float* scratch, void* stream) {
    if (!in1 || !gamma || !beta || !out || !scratch) {
        set_error("flowse_op_group_norm: null argument");
        return ERR_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!in2) C2 = 0;
    const int C = C1 + C2, HW = H * W;
    const int G = std::min(C / 4, 32);
    const int nblk = gn_partial_blocks(HW, C);
    float* part = scratch;
    float* mean = scratch + (int64_t)B * nblk * C * 2;
    float* scl = mean + (int64_t)B * C;
    int rc = launch_gn_stats(in1, C1, in2, C2, B, HW, part, nblk, s);
    if (rc != OK) return rc;
    rc = launch_gn_finalize(part, nblk, C, nullptr, 0, 0, B, HW, G, gamma, eps, mean, scl, s);
    if (rc != OK) return rc;
    GnParams p{mean, scl, beta};
    return launch_gn_apply(in1, C1, in2, C2, B, HW, p, silu, out, s);
}

int flowse_op_conv3x3_gn(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                         float eps, int silu, const float* w, const float* bias, const float* bias2, int bias2_stride,
                         const float* res, float* out, int B, int H, int W, int Cout, float scale, float* scratch,
                         void* stream) {
    if (!in1 || !gamma || !beta || !w || !out || !scratch) {
        set_error("flowse_op_conv3x3_gn: null argument");
        return ERR_ARG;
    }
    if (!in2) C2 = 0;
    if (!conv_supports_fused_gn(B, H, W, C1, C2, Cout, 9)) {
        set_error("flowse_op_conv3x3_gn: shape B=%d H=%d W=%d C=%d+%d Cout=%d not covered by the halo kernel", B, H, W,
                  C1, C2, Cout);
        return ERR_SHAPE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = C1 + C2, HW = H * W;
    const int G = std::min(C / 4, 32);
    const int nblk = gn_partial_blocks(HW, C);
    float* part = scratch;
    float* mean = scratch + (int64_t)B * nblk * C * 2;
    float* scl = mean + (int64_t)B * C;
    int rc = launch_gn_stats(in1, C1, in2, C2, B, HW, part, nblk, s);
    if (rc != OK) return rc;
    rc = launch_gn_finalize(part, nblk, C, nullptr, 0, 0, B, HW, G, gamma, eps, mean, scl, s);
    if (rc != OK) return rc;
    ConvArgs c;
    c.in1 = in1; c.in2 = in2; c.C1 = C1; c.C2 = C2;
    c.w = w; c.bias = bias; c.bias2 = bias2; c.bias2_stride = bias2_stride; c.res = res; c.out = out;
    c.B = B; c.H = H; c.W = W; c.Cout = Cout; c.taps = 9; c.scale = scale;
    c.gn = GnParams{mean, scl, beta};
    c.gn_silu = silu;
    return launch_conv(c, s);
}

static int op_conv3x3_winograd(int f43, const float* in1, int C1, const float* in2, int C2, const float* gamma,
                               const float* beta, float eps, int silu, const float* w, const float* bias,
                               const float* bias2, int bias2_stride, const float* res, float* out, int B, int H, int W,
                               int Cout, float scale, float* scratch, void* stream) {
    if (!in1 || !w || !out || !scratch || (gamma && !beta)) {
        set_error("flowse_op_conv3x3_f23/f43: null argument");
        return ERR_ARG;
    }
    if (!in2) C2 = 0;
    if (!conv_supports_wino(B, H, W, C1, C2, Cout, 9)) {
        set_error("flowse_op_conv3x3_f23/f43: shape B=%d H=%d W=%d C=%d+%d Cout=%d not covered by the Winograd kernel", B,
                  H, W, C1, C2, Cout);
        return ERR_SHAPE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = C1 + C2, HW = H * W;
    float* wf = scratch + flowse_op_group_norm_scratch_floats(B, HW, C);
    int rc = f43 ? launch_f43_weights(w, Cout, C, wf, s) : launch_wino_weights(w, Cout, C, wf, s);
    if (rc != OK) return rc;
    ConvArgs c;
    c.in1 = in1; c.in2 = in2; c.C1 = C1; c.C2 = C2;
    c.w = w; c.bias = bias; c.bias2 = bias2; c.bias2_stride = bias2_stride; c.res = res; c.out = out;
    c.B = B; c.H = H; c.W = W; c.Cout = Cout; c.taps = 9; c.scale = scale;
    c.wino = wf;
    c.wino_f43 = f43;
    const int ks = conv_ksplit(B, H, W, C, Cout, 9);
    if (f43 && ks > 1) {                           // same split plan as the model handle uses for this shape
        c.ksplit = ks;
        c.partial = wf + conv_wino_numel(Cout, C);
    }
    if (gamma) {
        const int G = std::min(C / 4, 32);
        const int nblk = gn_partial_blocks(HW, C);
        float* part = scratch;
        float* mean = scratch + (int64_t)B * nblk * C * 2;
        float* scl = mean + (int64_t)B * C;
        rc = launch_gn_stats(in1, C1, in2, C2, B, HW, part, nblk, s);
        if (rc != OK) return rc;
        rc = launch_gn_finalize(part, nblk, C, nullptr, 0, 0, B, HW, G, gamma, eps, mean, scl, s);
        if (rc != OK) return rc;
        c.gn = GnParams{mean, scl, beta};
        c.gn_silu = silu;
    }
    return launch_conv(c, s);
}

int64_t flowse_op_conv3x3_f23_scratch_floats(int B, int H, int W, int C, int Cout) {
    const int ks = conv_ksplit(B, H, W, C, Cout, 9);                   // > 1: F(4,3) runs split over K (small images)
    return flowse_op_group_norm_scratch_floats(B, H * W, C) + conv_wino_numel(Cout, C) +
           (ks > 1 ? (int64_t)ks * B * H * W * Cout : 0);
}

int flowse_op_conv3x3_f23(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                          float eps, int silu, const float* w, const float* bias, const float* bias2, int bias2_stride,
                          const float* res, float* out, int B, int H, int W, int Cout, float scale, float* scratch,
                          void* stream) {
    return op_conv3x3_winograd(0, in1, C1, in2, C2, gamma, beta, eps, silu, w, bias, bias2, bias2_stride, res, out, B, H,
                               W, Cout, scale, scratch, stream);
}
int flowse_op_conv3x3_f43(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                          float eps, int silu, const float* w, const float* bias, const float* bias2, int bias2_stride,
                          const float* res, float* out, int B, int H, int W, int Cout, float scale, float* scratch,
                          void* stream) {
    return op_conv3x3_winograd(1, in1, C1, in2, C2, gamma, beta, eps, silu, w, bias, bias2, bias2_stride, res, out, B, H,
                               W, Cout, scale, scratch, stream);
}

// 16-bit storage per-op entry: fp32 NHWC tensors at the boundary, rounded to bf16 (dt 1) / half (dt 2) inside, conv on
// the 16-bit matrix cores (LDS-halo kernel when it applies, else the flat kernel), result widened back.  Optional fused
// GroupNorm(+SiLU) on the input (halo shapes only) from caller-supplied per-(sample, channel) mean / scale and beta.
int flowse_op_conv2d_16(const float* in1, int C1, const float* in2, int C2, const float* w, const float* bias,
                        const float* res, const float* gn_mean, const float* gn_scale, const float* gn_beta, int silu,
                        float* out, int B, int H, int W, int Cout, int taps, float scale, int dt, void* scratch,
                        int64_t scratch_bytes, void* stream) {
    if (!in1 || !w || !out || !scratch || (dt != DT_BF16 && dt != DT_F16) || (taps != 1 && taps != 9)) {
        set_error("flowse_op_conv2d_16: bad argument");
        return ERR_ARG;
    }
    if (!in2) C2 = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t M = (int64_t)B * H * W, C = C1 + C2;
    const bool halo = conv16_uses_halo(B, H, W, C1, C2, Cout, taps);
    const int ks = halo ? 1 : conv16_ksplit(B, H, W, (int)C, Cout, taps);
    const int64_t nw = ((int64_t)Cout * taps * C + 3) & ~(int64_t)3;
    int64_t need = 2 * (M * C1 + M * C2 + nw + 2 * M * Cout) + 64 + (ks > 1 ? 4 * (int64_t)ks * M * Cout : 0);
    if (scratch_bytes < need + 256) {
        set_error("flowse_op_conv2d_16: scratch needs %lld bytes", (long long)(need + 256));
        return ERR_ARG;
    }
    if (gn_mean && !halo) {
        set_error("flowse_op_conv2d_16: fused GroupNorm input only on LDS-halo shapes");
        return ERR_SHAPE;
    }
    char* p = static_cast<char*>(scratch);
    auto take = [&](int64_t bytes) { char* q = p; p += (bytes + 255) & ~(int64_t)255; return q; };
    void* a1 = take(2 * M * C1);
    void* a2 = C2 ? take(2 * M * C2) : nullptr;
    void* wq = take(2 * nw);
    void* r16 = res ? take(2 * M * Cout) : nullptr;
    void* o16 = take(2 * M * Cout);
    float* part = ks > 1 ? reinterpret_cast<float*>(take(4 * (int64_t)ks * M * Cout)) : nullptr;
    if ((size_t)(p - static_cast<char*>(scratch)) > (size_t)scratch_bytes) {
        set_error("flowse_op_conv2d_16: scratch too small");
        return ERR_ARG;
    }
    int rc = launch_convert(in1, DT_F32, a1, dt, M * C1, s);
    if (rc == OK && C2) rc = launch_convert(in2, DT_F32, a2, dt, M * C2, s);
    if (rc == OK) rc = launch_convert(w, DT_F32, wq, dt, nw, s);
    if (rc == OK && res) rc = launch_convert(res, DT_F32, r16, dt, M * Cout, s);
    if (rc != OK) return rc;
    ConvArgs c;
    c.in1 = static_cast<const float*>(a1); c.in2 = static_cast<const float*>(a2); c.C1 = C1; c.C2 = C2;
    c.w = w; c.bias = bias; c.bias2 = nullptr; c.bias2_stride = 0;
    c.res = static_cast<const float*>(r16); c.out = static_cast<float*>(o16);
    c.B = B; c.H = H; c.W = W; c.Cout = Cout; c.taps = taps; c.scale = scale;
    c.ksplit = ks; c.partial = part;
    c.wq = wq; c.terms = 1; c.wq_f16 = dt == DT_F16 ? 1 : 0;
    c.in_dt = dt; c.out_dt = dt;
    if (gn_mean) {
        c.gn = GnParams{gn_mean, gn_scale, gn_beta};
        c.gn_silu = silu;
    }
    rc = launch_conv(c, s);
    if (rc != OK) return rc;
    return launch_convert(o16, dt, out, DT_F32, M * Cout, s);
}

int flowse_op_fir_up(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    GnParams p{nullptr, nullptr, nullptr};
    return launch_fir_up(in, B, H, W, C, p, 0, nullptr, out, static_cast<hipStream_t>(stream));
}
int flowse_op_fir_down(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    GnParams p{nullptr, nullptr, nullptr};
    return launch_fir_down(in, B, H, W, C, p, 0, out, static_cast<hipStream_t>(stream));
}
int flowse_op_attention(const float* qkv, float* out, int B, int L, int C, void* stream) {
    return launch_attention(qkv, B, L, C, out, static_cast<hipStream_t>(stream));
}
int flowse_op_gfp(const float* t, const float* W, float* out, int B, int E, void* stream) {
    return launch_gfp(t, W, B, E, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
