// Model handle, part 1: error reporting, NCSN++ module list / parameter table (mirrors NCSNpp.__init__,
// flowmse/backbones/ncsnpp.py:97-245: module order = parameter order) and the weight packer.
#include "model.h"

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

static bool in_list(const int32_t* v, int n, int x) {
    for (int i = 0; i < n; ++i)
        if (v[i] == x) return true;
    return false;
}

// 16-bit storage applies when every wide tensor of the network has a multiple of 32 channels (what the 16-bit
// matrix-core kernels tile by); otherwise precision 2 / 3 only switch the operands of the big 3x3 convs (fp32 storage).
int storage_type_for(const flowse_model* m) {
    if (m->precision < 2) return DT_F32;
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

void add_module(flowse_model* m, Module mod) {
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

Module resblock_module(int in_ch, int out_ch, bool up, bool down) {
    Module r;
    r.kind = M_RESBLOCK;
    r.in_ch = in_ch;
    r.out_ch = out_ch;
    r.up = up;
    r.down = down;
    r.shortcut = (in_ch != out_ch) || up || down;     // layerspp.py:234-235
    return r;
}
Module simple_module(ModKind k, int in_ch, int out_ch) {
    Module r;
    r.kind = k;
    r.in_ch = in_ch;
    r.out_ch = out_ch;
    return r;
}

// NCSNpp.__init__, ncsnpp.py:97-245
int build_structure(flowse_model* m) {
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

    add_module(m, simple_module(M_GFP, 0, nf));
    add_module(m, simple_module(M_LINEAR, 2 * nf, 4 * nf));
    add_module(m, simple_module(M_LINEAR, 4 * nf, 4 * nf));
    add_module(m, simple_module(M_CONV3, 4, nf));
    std::vector<int> hs_c{nf};
    int in_ch = nf;
    for (int lv = 0; lv < L; ++lv) {
        const int res = c.image_size >> lv;
        for (int b = 0; b < c.num_res_blocks; ++b) {
            const int out_ch = nf * c.ch_mult[lv];
            add_module(m, resblock_module(in_ch, out_ch));
            in_ch = out_ch;
            if (in_list(c.attn_resolutions, c.num_attn, res)) add_module(m, simple_module(M_ATTN, in_ch, in_ch));
            hs_c.push_back(in_ch);
        }
        if (lv != L - 1) {
            add_module(m, resblock_module(in_ch, in_ch, false, true));
            add_module(m, simple_module(M_COMBINE, 4, in_ch));
            hs_c.push_back(in_ch);
        }
    }
    in_ch = hs_c.back();
    add_module(m, resblock_module(in_ch, in_ch));
    add_module(m, simple_module(M_ATTN, in_ch, in_ch));
    add_module(m, resblock_module(in_ch, in_ch));
    for (int lv = L - 1; lv >= 0; --lv) {
        const int res = c.image_size >> lv;
        for (int b = 0; b < c.num_res_blocks + 1; ++b) {
            const int out_ch = nf * c.ch_mult[lv];
            add_module(m, resblock_module(in_ch + hs_c.back(), out_ch));
            hs_c.pop_back();
            in_ch = out_ch;
        }
        if (in_list(c.attn_resolutions, c.num_attn, res)) add_module(m, simple_module(M_ATTN, in_ch, in_ch));
        add_module(m, simple_module(M_GN, in_ch, in_ch));
        add_module(m, simple_module(M_CONV3, in_ch, 4));
        if (lv != 0) add_module(m, resblock_module(in_ch, in_ch, true, false));
    }
    if (!hs_c.empty()) {
        set_error("internal: skip stack not empty");
        return ERR_STATE;
    }
    return OK;
}

// conv weight [Cout][Cin][kh][kw] -> [Cout][kh*kw][Cin]
static int64_t pack_conv(Packer& pk, const float* src, int Cout, int Cin, int taps) {
    const int64_t off = pk.put((int64_t)Cout * taps * Cin);
    float* dst = pk.host.data() + off;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < taps; ++t)
                dst[((int64_t)co * taps + t) * Cin + ci] = src[((int64_t)co * Cin + ci) * taps + t];
    if (taps == 9 && (Cin % 32) == 0 && (Cout % 64) == 0) pk.wino.push_back({off, Cout, Cin});
    if ((Cin % 32) == 0 && (Cout % 32) == 0) pk.smallm.push_back({off, Cout, Cin, taps});
    return off;
}
static int64_t pack_copy(Packer& pk, const float* src, int64_t n) {
    const int64_t off = pk.put(n);
    memcpy(pk.host.data() + off, src, n * sizeof(float));
    return off;
}

int pack_weights(flowse_model* m, const float* blob, Packer& pk) {
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
                // the two projections are 1x1 convs: fragment-order copies for the small-M kernel (single utterances)
                if ((C % 32) == 0) {
                    pk.smallm.push_back({mod.w_qkv, 3 * C, C, 1});
                    pk.smallm.push_back({mod.w_o, C, C, 1});
                }
                break;
            }
        }
    }
    pk.host.resize((pk.host.size() + 63) & ~(size_t)63, 0.f);      // whole float4s (the 16-bit twin converts by quads)
    return OK;
}

}  // namespace flowse

extern "C" const char* flowse_last_error(void) { return flowse::g_err; }
