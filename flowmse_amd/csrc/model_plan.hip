// Model handle, part 2: the launch plan.  Mirrors NCSNpp.forward (flowmse/backbones/ncsnpp.py:247-404: the order the
// modules are consumed in).  The forward pass is "traced" once per input shape into a flat list of kernel launches over
// an arena-planned activation workspace: no allocation, no host synchronisation and no shape logic inside the N-step
// solver loop.
#include "model.h"

namespace flowse {

static bool in_list(const int32_t* v, int n, int x) {
    for (int i = 0; i < n; ++i)
        if (v[i] == x) return true;
    return false;
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
// The 1x1 shortcut Conv_2(x) of a ResnetBlock folded into Conv_1's launch (ConvArgs::sc1, conv3x3_pc16_kernel only)
struct ScFold {
    const Tn* s1 = nullptr;
    const Tn* s2 = nullptr;             // second part of a concat input, or null
    int64_t w = -1, bias = -1;          // packed offsets of Conv_2's weight / bias
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
    bool failed = false;                   // an internal planning inconsistency: build_plan returns ERR_STATE
    void op(const std::string& label, std::function<int(hipStream_t)> f, double flops = 0.0, double bytes = 0.0,
            bool dominant = false, double issued = -1.0) {
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
            SkPartial* defer = nullptr, const SkPartial* extra = nullptr, GnFuse* gnf = nullptr,
            const ScFold* fold = nullptr, bool no_stats = false) {
        flowse_model* M = m;
        const int C1 = a.C, C2 = b2 ? b2->C : 0, H = a.H, Wd = a.W, Bn = B;
        {   // a deferred reduction leaves no output tensor: decide before anything is allocated
            const bool in16_ = a.dt != DT_F32;
            const int ks_ = cin4 ? 1 : in16_ ? ((conv_supports_head4(Bn, H, Wd, C1, C2, Cout, taps) || conv16_uses_halo(Bn, H, Wd, C1, C2, Cout, taps))
                                                    ? 1 : conv16_ksplit(Bn, H, Wd, C1 + C2, Cout, taps))
                                             : conv_ksplit(Bn, H, Wd, C1 + C2, Cout, taps);
            if (defer && !(ks_ > 1 && !res && dense_row0 < 0)) defer = nullptr;
            if ((extra || gnf) && !(ks_ > 1)) {
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
        // Combine (conv1x1 4 -> C, in place on `res`, any storage type): the kernel leaves the statistics of the updated tensor
        if (cin4 && out_is_res && taps == 1 && C1 == 4 && !has2 && idt == DT_F32 && Cout >= 16)
            st_nblk = conv_cin4_stats_blocks(Bn, H, Wd, Cout);
        // conv_ksplit / conv_fused_stats_blocks judge the small-M kernel by the TOTAL channel count; a concat whose parts are
        // not 32-aligned (not in the released net) runs the unsplit flat kernel instead: its statistics geometry applies
        const bool use_smallm = !in16 && !cin4 && M->wsm_offs.count(w) && conv_smallm_ok(Bn, H, Wd, C1, C2, Cout, taps);
        if (!in16 && !cin4 && !use_smallm && st_nblk > 0 && conv_smallm_ok(Bn, H, Wd, C1 + C2, 0, Cout, taps))
            st_nblk = (H * Wd) % 128 == 0 ? H * Wd / 128 : 0;
        const bool use_stream = !in16 && !cin4 && ks == 1 && M->wsm_offs.count(w) && conv1x1_stream_ok(Bn, H, Wd, C1, C2, Cout, taps);
        if (!in16 && !cin4 && !use_stream && st_nblk > 0 && conv1x1_stream_ok(Bn, H, Wd, C1 + C2, 0, Cout, taps))
            st_nblk = (H * Wd) % 128 == 0 ? H * Wd / 128 : 0;
        if (in16 && !cin4 && st_nblk > 0 && conv16_smallm_ok(Bn, H, Wd, C1 + C2, 0, Cout, taps) &&
            !(M->frag_offs.count(w) && conv16_smallm_ok(Bn, H, Wd, C1, C2, Cout, taps)))
            st_nblk = (H * Wd) % 128 == 0 ? H * Wd / 128 : 0;
        // the two-dimensional Winograd kernel takes this launch: statistics in 4 x 16 pixel strips
        const auto w2_it = M->wino2_of.find(w);
        const int64_t wino2_off = (!in16 && taps == 9 && !cin4 && ks == 1 && w2_it != M->wino2_of.end() &&
                                   !(wq_off >= 0 && M->precision != 0) && conv_supports_fused_gn(Bn, H, Wd, C1, C2, Cout, taps) &&
                                   conv_supports_w2d(Bn, H, Wd, C1, C2, Cout, taps)) ? w2_it->second : -1;
        if (wino2_off >= 0 && st_nblk > 0) st_nblk = H * Wd / 64;
        if (defer || gnf) st_nblk = 0;                   // no output here / the statistics are finished inside the reduction
        if (no_stats) st_nblk = 0;                       // nobody normalises this tensor (the shortcut: a residual only)
        if (out_is_res && o.st_nblk > 0) {               // updated in place: the producer's statistics are stale
            arena.release(o.st_off);
            o.st_nblk = 0;
        }
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
        const size_t table_off = M_table_off;     // by value: the Builder dies before the plan runs
        const bool has_extra = extra != nullptr && extra->valid;
        const SkPartial xp = has_extra ? *extra : SkPartial();
        const bool has_fold = fold != nullptr;
        const size_t f1_off = has_fold ? fold->s1->off : 0, f2_off = has_fold && fold->s2 ? fold->s2->off : 0;
        const int FC1 = has_fold ? fold->s1->C : 0, FC2 = has_fold && fold->s2 ? fold->s2->C : 0;
        const int64_t fold_w = has_fold ? fold->w : -1, fold_b = has_fold ? fold->bias : -1;
        if (has_fold && (res || extra || ks != 1 || !in16 || taps != 9 || fold->s1->H != H || fold->s1->W != Wd ||
                         fold->s1->dt != idt || !M->frag_offs.count(w) || !M->frag_offs.count(fold_w))) {
            set_error("internal: shortcut fold requested for a conv that cannot take it (%s)", label.c_str());
            failed = true;
        }
        const std::string full_label = label + "@" + std::to_string(H) + "x" + std::to_string(Wd) + ":" +
                                       std::to_string(C1 + C2) + (has_fold ? "+" + std::to_string(FC1 + FC2) : std::string()) +
                                       ">" + std::to_string(Cout);
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
            c.stats = st_nblk > 0 ? M->A(st_off) : nullptr;
            c.stats_nblk = st_nblk;
            if (has_gin) {
                c.gn = GnParams{M->A(gbuf.mean), M->A(gbuf.scale), M->W(gbuf.beta)};
                c.gn_silu = gin_silu ? 1 : 0;
            }
            if (has_fold) {                               // Conv_2(x) as extra K steps of this launch
                c.sc1 = M->A(f1_off);
                c.SC1 = FC1;
                c.sc2 = FC2 ? M->A(f2_off) : nullptr;
                c.SC2 = FC2;
                c.wfrag_sc = M->d_wfrag + fold_w;
                c.bias_x = fold_b >= 0 ? M->W(fold_b) : nullptr;
            }
            c.in_dt = idt;
            c.out_dt = odt;
            if (in16) {                                   // [Cout][taps][Cin] in the storage type: same offsets as d_w
                c.wq = M->d_w16 + w;
                if (M->frag_offs.count(w)) c.wfrag = M->d_wfrag + w;
                c.terms = 1;
                c.wq_f16 = idt == DT_F16 ? 1 : 0;
            } else if (use_bf16) {
                c.wq = M->d_wq + wq_off;
                c.terms = terms;
                c.wq_f16 = M->precision == 3 ? 1 : 0;
            }
            if (wino_off >= 0) c.wino = M->d_wino + wino_off;
            if (wino2_off >= 0) c.wino2 = M->d_wino2 + wino2_off;
            if (use_smallm) {
                c.wsm = M->d_wsm + w;
                c.wsm16 = M->d_wsm16 + w;
            }
            if (use_stream) c.wsm = M->d_wsm + w;
            return c;
        };
        const double flops = 2.0 * Bn * H * Wd * (double)Cout * (taps * (C1 + C2) + (FC1 + FC2));
        const double out_bytes = (double)dt_size(odt) * Bn * H * Wd * Cout * (hasres ? 2 : 1);
        const double in_bytes = (double)dt_size(idt) * ((double)Bn * H * Wd * (C1 + C2 + FC1 + FC2) +
                                                        (double)Cout * (taps * (C1 + C2) + (FC1 + FC2)));
        const double part_bytes = 4.0 * ks * (double)Bn * H * Wd * Cout;
        op(full_label, [=](hipStream_t s) {
            const ConvArgs c = make_args();
            return cin4 ? launch_conv_cin4(c, s) : launch_conv(c, s, false);
        }, flops, in_bytes + (ks > 1 ? part_bytes : out_bytes),
           has_gin && Cout > 64 && ks == 1 && (wino2_off >= 0 || wino_off < 0 || conv_f43_wide(Bn, H, Wd, Cout)),
           wino2_off >= 0 ? flops / 3.0 : wino_off >= 0 ? flops * 0.5 : (use_bf16 && terms == 3) ? 3.0 * flops : flops);
        if (defer) {                                     // the consumer's reduction sums these slices (ConvArgs::partial2)
            defer->part_off = part_off;
            defer->ks = ks;
            defer->bias = bias;
            defer->valid = true;
            return Tn();
        }
        if (ks > 1 && gnf) {
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
        } else if (ks > 1)
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

    // A deferred split-K conv whose consumer turned out not to run a two-pass reduction after all: its own reduction launch
    Tn reduce_deferred(SkPartial& sp, int H, int Wd, int Cout) {
        flowse_model* M = m;
        Tn o = alloc(H, Wd, Cout);
        const size_t part = sp.part_off, o_off = o.off;
        const int ks = sp.ks, Bn = B, odt = o.dt;
        const int64_t bias = sp.bias;
        op("splitk_reduce@" + std::to_string(H) + "x" + std::to_string(Wd), [=](hipStream_t s) {
            ConvArgs c;
            c.B = Bn; c.H = H; c.W = Wd; c.Cout = Cout; c.C1 = Cout; c.taps = 1;
            c.scale = 1.f;
            c.ksplit = ks;
            c.partial = M->A(part);
            c.bias = bias >= 0 ? M->W(bias) : nullptr;
            c.out = M->A(o_off);
            c.in_dt = odt;
            c.out_dt = odt;
            return launch_splitk_reduce(c, s);
        }, 0.0, 4.0 * ks * (double)Bn * H * Wd * Cout + (double)dt_size(odt) * Bn * H * Wd * Cout);
        arena.release(sp.part_off);
        sp.valid = false;
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
        // 16-bit storage, Conv_1 on the producer / consumer kernel: the shortcut Conv_2(x) runs as extra K steps of Conv_1's
        // launch (ConvArgs::sc1) -- no launch, no round trip of its output through HBM, one read of x less
        const int xc2 = (!mod.up && !mod.down && x2) ? x2->C : 0;
        const bool fold_sc = mod.shortcut && m->act_dt != DT_F32 && !merge_sc && !getenv("FLOWSE_NO_SCFOLD") &&
                             m->frag_offs.count(mod.w_c1) && m->frag_offs.count(mod.w_c2) &&
                             conv16_uses_pc(B, Ho, Wo, mod.out_ch, 0, mod.out_ch, 9) &&
                             fusable_shape(m->act_dt, Ho, Wo, mod.out_ch, 0) && (x1.C % 32) == 0 && (xc2 % 32) == 0 &&
                             x1.C + xc2 >= 96 && x1.dt == m->act_dt;
        Tn xr;
        if (!mod.up && !mod.down) {
            if (mod.shortcut && !fold_sc) {
                xs = conv("conv2_1x1", x1, x2, mod.w_c2, mod.w_c2_b, -1, mod.out_ch, 1, nullptr, 1.f, false, false, nullptr,
                          false, -1, -1, merge_sc ? &sp : nullptr, nullptr, nullptr, nullptr, true);
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
            Tn hr = fir(x1, mod.up, &g0, true, nullptr, false, &xr);      // act(GN(x)) and x resampled in one pass
            gn_release(g0);
            // the shortcut Conv_2(x) (layerspp.py:268-270)
            if (!fold_sc)
                xs = conv("conv2_1x1", xr, nullptr, mod.w_c2, mod.w_c2_b, -1, mod.out_ch, 1, nullptr, 1.f, false, false, nullptr,
                          false, -1, -1, merge_sc ? &sp : nullptr, nullptr, nullptr, nullptr, true);
            h1 = conv("conv0_3x3", hr, nullptr, mod.w_c0, -1, mod.dense_row0, mod.out_ch, 9, nullptr, 1.f, false, false,
                      nullptr, false, mod.wq_c0, -1, nullptr, nullptr, &gf);
            release(hr);
            if (!fold_sc) release(xr);
        }
        Tn out;
        // The two fusions above were requested from PREDICTED shapes / types; Conv_1 is decided from the tensors that exist:
        //  * the shortcut's slices can only ride on Conv_1's reduction if Conv_1 itself runs split -- else they get their
        //    own reduction launch now;
        //  * per-channel mean / scale from Conv_0's reduction are only of use to a Conv_1 that normalises on load -- else
        //    they are released and GroupNorm_1 is materialised below.
        if (sp.valid && !sk_two_pass(h1.dt, h1.H, h1.W, h1.C, mod.out_ch, 9)) xs = reduce_deferred(sp, h1.H, h1.W, mod.out_ch);
        if (gf.done && !gf.apply && !fusable(h1, 0)) {
            gn_release(gf.g);
            gf.done = false;
        }
        const Tn* resid = (sp.valid || fold_sc) ? nullptr : (xs.valid() ? &xs : &x1);
        const SkPartial* extra = sp.valid ? &sp : nullptr;
        ScFold fo;
        if (fold_sc) {
            fo.s1 = (mod.up || mod.down) ? &xr : &x1;
            fo.s2 = (mod.up || mod.down) ? nullptr : x2;
            fo.w = mod.w_c2;
            fo.bias = mod.w_c2_b;
        }
        const ScFold* fold = fold_sc ? &fo : nullptr;
        if (fold && ((gf.done && gf.apply) || !fusable(h1, 0) || !conv16_uses_pc(B, h1.H, h1.W, h1.C, 0, mod.out_ch, 9) ||
                     getenv("FLOWSE_SCFOLD_LATE"))) {          // (the variable: test hook that forces this branch)
            // The fold was planned from PREDICTED shapes / types and the shortcut launch skipped; the Conv_1 that exists does not
            // take it (the two predicates agree today: this is the safety net for a policy that drifts).  The shortcut runs
            // now, as its own launch, and rides as Conv_1's residual: a little slower, never an unusable model.
            xs = conv("conv2_1x1", *fo.s1, fo.s2, mod.w_c2, mod.w_c2_b, -1, mod.out_ch, 1, nullptr, 1.f, false, false, nullptr,
                      false, -1, -1, nullptr, nullptr, nullptr, nullptr, true);
            if (mod.up || mod.down) release(xr);
            resid = &xs;
            fold = nullptr;
        }
        if (gf.done && gf.apply) {                       // h1 already is act(GroupNorm_1(Conv_0(.)))
            out = conv("conv1_3x3", h1, nullptr, mod.w_c1, mod.w_c1_b, -1, mod.out_ch, 9, resid, rs2, false, false, nullptr,
                       false, -1, -1, nullptr, extra);
            release(h1);
        } else if (fusable(h1, 0)) {
            GnBuf g1 = gf.done ? gf.g : gn(h1, nullptr, mod.w_gn1_g, mod.w_gn1_b);
            out = conv(fold ? "conv1_3x3_gn_sc" : "conv1_3x3_gn", h1, nullptr, mod.w_c1, mod.w_c1_b, -1, mod.out_ch, 9, resid, rs2,
                       false, false, &g1, true, mod.wq_c1, -1, nullptr, extra, nullptr, fold);
            gn_release(g1);
            release(h1);
            if (fold && (mod.up || mod.down)) release(xr);
        } else {
            Tn h2 = gn_norm(h1, nullptr, mod.w_gn1_g, mod.w_gn1_b, true);
            release(h1);
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
        // 16-bit storage modes: the whole sub-block stays in the activation type -- GroupNorm rounds to it, the qkv
        // projection and the output projection run the 16-bit flat kernel, the attention core the 16-bit MFMA kernel
        // (fp32 softmax state / accumulation); fp32 mode: everything fp32
        const int adt = x.dt;
        Tn hn = gn_norm(x, nullptr, mod.w_gn0_g, mod.w_gn0_b, false, adt);
        Tn qkv = conv("attn_qkv", hn, nullptr, mod.w_qkv, mod.w_qkv_b, -1, 3 * C, 1, nullptr, 1.f, false, false, nullptr,
                      false, -1, adt);
        release(hn);
        Tn o = alloc(x.H, x.W, C, adt);
        const size_t q_off = qkv.off, o_off = o.off;
        op("attention@" + std::to_string(x.H) + "x" + std::to_string(x.W), [=](hipStream_t s) { return launch_attention(M->A(q_off), Bn, L, C, M->A(o_off), s, adt); },
           4.0 * Bn * (double)L * L * C, 4.0 * dt_size(adt) * Bn * L * C);
        release(qkv);
        Tn out = conv("attn_out", o, nullptr, mod.w_o, mod.w_o_b, -1, C, 1, &x, rs2, false, false, nullptr, false, -1,
                      x.dt);
        release(o);
        return out;
    }
};

// NCSNpp.forward, ncsnpp.py:247-404
int build_plan(flowse_model* m, Plan* plan, int B, int F, int T) {
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
            h = bd.conv("combine_1x1", ipyr, nullptr, cb.w_a, cb.w_a_b, -1, cb.out_ch, 1, &h, 1.f, true, true);
            hs.push_back(h);           // (updated in place; its statistics are the Combine kernel's, or dropped)
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

// Plan of a single-module handle: inputs are copied into the arena, the module runs exactly as inside the network
// (Builder::resblock / attn / conv), the result is copied out.
int build_block_plan(flowse_model* m, Plan* plan, int B, int H, int W, int C1) {
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

}  // namespace flowse
