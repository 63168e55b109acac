// Internal types of the model handle: NCSN++ module list, parameter table, launch plan (model_*.hip).
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/flowse_hip.h"
#include "common.h"

namespace flowse {

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
    void release(size_t off) {
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
};

struct Plan {
    int B = 0, F = 0, T = 0;
    size_t ws_bytes = 0;
    std::vector<std::function<int(hipStream_t)>> ops;
    std::vector<std::string> labels;
    std::vector<double> flops, bytes;      // algorithmic work / HBM traffic of each launch
    std::vector<double> issued;            // FLOPs the matrix cores execute for it (Winograd forms: 1/2 or 2/3 of `flops`)
    std::vector<char> dominant;            // 1 = a launch of the dominant kernel: the unsplit 3x3 ResBlock conv with fused
                                           // GroupNorm+SiLU input (conv3x3_f43_kernel<2, false, 2> in the fp32 mode)
    // The launch list holds no per-call argument (those live in the handle's device-resident CallBlock): by default it is
    // issued as plain launches; under FLOWSE_GRAPH=1 it is captured after one eager pass and replayed as a hipGraph.
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
    // 3x3 convs with Cin % 32 == 0 and Cout % 128 == 0 again in MFMA fragment order (launch_pc16_weights) at the SAME offsets
    // as in d_w16: the B operand of conv3x3_pc16_kernel's consumer waves, straight from L2
    uint16_t* d_wfrag = nullptr;
    int64_t d_wfrag_numel = 0;
    std::set<int64_t> frag_offs;           // packed weight offsets that have fragment-order weights
    float* d_wino = nullptr;               // F(4,3) Winograd weights of the 3x3 convs the Winograd kernel can take
    int64_t d_wino_numel = 0;
    std::map<int64_t, int64_t> wino_of;    // packed weight offset (d_w) -> offset in d_wino
    float* d_wino2 = nullptr;              // F(4,3) x F(2,3) weights of the same convs (conv3x3_w2d_kernel; FLOWSE_W2D=0: absent)
    int64_t d_wino2_numel = 0;
    std::map<int64_t, int64_t> wino2_of;
    float* d_wsm16 = nullptr;              // the same in the 16 x 16-tile fragment order (conv_smallm16_kernel), same offsets
    float* d_wsm = nullptr;                // fragment-order copy of the conv weights with 32-aligned channel counts (conv_smallm_kernel)
    int64_t d_wsm_numel = 0;
    std::set<int64_t> wsm_offs;            // packed weight offsets (d_w) that have a copy at the SAME offset in d_wsm
    char* d_ws = nullptr;                  // activation workspace
    size_t d_ws_bytes = 0;
    float* d_ts = nullptr;                 // [N][B] solver times
    size_t d_ts_floats = 0;
    std::map<std::tuple<int, int, int>, Plan> plans;
    CallBlock* d_call = nullptr;           // per-call arguments of the boundary kernels, in device memory
    int device = -1;                       // HIP device that owns every d_* buffer of this handle
    bool use_graph = false;                // FLOWSE_GRAPH=1: replay each shape's launch list as a hipGraph (slower, measured)
    // Callers on the NULL (legacy default) stream -- PyTorch's default stream IS the NULL stream -- cannot be captured;
    // their work runs on this internal stream instead, fenced against the NULL stream by events on both sides.
    hipStream_t gstream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int64_t graph_launches = 0;            // hipGraphLaunch calls made by this handle (flowse_model_graph_launches)
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

// model_build.hip
int build_structure(flowse_model* m);
void add_module(flowse_model* m, Module mod);
Module resblock_module(int in_ch, int out_ch, bool up = false, bool down = false);
Module simple_module(ModKind k, int in_ch, int out_ch);
struct Packer {
    std::vector<float> host;
    struct WinoReq { int64_t off; int Cout, Cin; };
    std::vector<WinoReq> wino;             // 3x3 convs that also get F(4,3) weights (transformed on the device)
    struct SmReq { int64_t off; int Cout, Cin, taps; };
    std::vector<SmReq> smallm;             // convs that also get a fragment-order copy (small-M kernel)
    int64_t put(int64_t n) {
        const int64_t off = ((int64_t)host.size() + 63) & ~(int64_t)63;
        host.resize(off + n, 0.f);
        return off;
    }
};
int pack_weights(flowse_model* m, const float* blob, Packer& pk);
int storage_type_for(const flowse_model* m);
// model_plan.hip
int build_plan(flowse_model* m, Plan* plan, int B, int F, int T);
int build_block_plan(flowse_model* m, Plan* plan, int B, int H, int W, int C1);

}  // namespace flowse
