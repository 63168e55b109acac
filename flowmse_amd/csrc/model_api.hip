// Model handle, part 3: plan execution (plain launches, optional hipGraph replay), device state and the C ABI
// (include/flowse_hip.h).
#include "model.h"

namespace flowse {

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
    for (size_t i = 0; i < p->ops.size(); ++i) {
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
        const int rc = p->ops[i](s);
        if (rc != OK) return rc;
        if (prof) {
            FLOWSE_HIP(hipEventRecord(pd.b, s));
            m->prof_pending.push_back(pd);
        }
    }
    return OK;
}

// One network evaluation.  Default: plain launches of the shape's launch list.  FLOWSE_GRAPH=1 (opt-in: measured 5 %
// slower than plain launches at batch 1, equal at batch 8): first call per shape eager (also performs the one-time
// per-device kernel attribute setup), second call captures the same launch list into a hipGraph, afterwards one
// hipGraphLaunch per call.  Profiling (per-launch events) always runs plain launches.
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
    if (caller != nullptr || !m->use_graph || m->prof_mode != -1) return OK;
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
    if (m->d_wfrag) (void)hipFree(m->d_wfrag);
    if (m->d_wino) (void)hipFree(m->d_wino);
    if (m->d_wino2) (void)hipFree(m->d_wino2);
    m->d_wino2 = nullptr;
    m->d_wino2_numel = 0;
    if (m->d_wsm) (void)hipFree(m->d_wsm);
    if (m->d_wsm16) (void)hipFree(m->d_wsm16);
    m->d_wsm16 = nullptr;
    m->d_wsm = nullptr;
    m->d_wsm_numel = 0;
    if (m->d_call) (void)hipFree(m->d_call);
    if (m->d_rk) (void)hipFree(m->d_rk);
    m->d_rk = nullptr;
    m->d_rk_floats = 0;
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
    m->d_wfrag = nullptr;
    m->d_w_numel = m->d_wq_numel = m->d_wino_numel = m->d_w16_numel = m->d_wfrag_numel = 0;
    m->d_ws_bytes = m->d_ts_floats = 0;
    m->device = -1;
    if (sw) (void)hipSetDevice(cur);
}

}  // namespace flowse
// =============================================================================================== C ABI
extern "C" {

int flowse_abi_version(void) { return FLOWSE_ABI_VERSION; }

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
    // calls per 6 ms network evaluation at batch 1).
    m->use_graph = getenv("FLOWSE_GRAPH") != nullptr;
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
    m->block_kind = kind;
    m->temb_dim = temb_dim;
    if (kind == FLOWSE_BLOCK_RESNET) add_module(m, resblock_module(in_ch, out_ch, up != 0, down != 0));
    else add_module(m, simple_module(kind == FLOWSE_BLOCK_ATTN ? M_ATTN : M_COMBINE, in_ch, out_ch));
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
        // fragment-order copies for the producer / consumer 3x3 kernel, same offsets as in d_w16
        m->frag_offs.clear();
        if (m->d_wfrag && m->d_wfrag_numel < n16) {
            FLOWSE_HIP(hipFree(m->d_wfrag));
            m->d_wfrag = nullptr;
        }
        if (!m->d_wfrag) {
            FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_wfrag), n16 * sizeof(uint16_t)));
            m->d_wfrag_numel = n16;
        }
        for (auto& r : pk.wino) {
            if ((r.Cout % 128) != 0 || (int64_t)r.Cout * 9 * r.Cin * 2 >= (1LL << 31)) continue;
            const int frc = launch_pc16_weights(m->d_w16 + r.off, r.Cout, r.Cin, m->d_wfrag + r.off, nullptr);
            if (frc != OK) return frc;
            m->frag_offs.insert(r.off);
        }
        // ... and for every other conv with 32-aligned channel counts (1x1 shortcuts, attention projections, 3x3 with
        // Cout % 128 != 0): the 16-bit small-image kernel reads the same layout
        if (conv16_smallm_ok(1, 4, 4, 32, 0, 32, 1)) {
            for (auto& r : pk.smallm) {
                if (m->frag_offs.count(r.off) || (int64_t)r.Cout * r.taps * r.Cin * 2 >= (1LL << 31)) continue;
                const int frc = launch_pc16_weights(m->d_w16 + r.off, r.Cout, r.Cin, m->d_wfrag + r.off, nullptr, r.taps);
                if (frc != OK) return frc;
                m->frag_offs.insert(r.off);
            }
        }
        pk.wino.clear();                     // no fp32 Winograd kernels run on 16-bit activations
    }
    // F(4,3) Winograd weights, derived on the device from the packed fp32 weights just uploaded
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
        const int wrc = launch_f43_weights(m->d_w + r.off, r.Cout, r.Cin, dst, nullptr);
        if (wrc != OK) return wrc;
    }
    // fragment-order copies for the small-M kernel (fp32 activations only), at the same offsets as in d_w
    m->wsm_offs.clear();
    if (!m->storage16() && !pk.smallm.empty() && conv_smallm_ok(1, 4, 4, 32, 0, 32, 1)) {
        const int64_t nw = (int64_t)pk.host.size();
        if (m->d_wsm && m->d_wsm_numel < nw) {
            FLOWSE_HIP(hipFree(m->d_wsm));
            FLOWSE_HIP(hipFree(m->d_wsm16));
            m->d_wsm = m->d_wsm16 = nullptr;
        }
        if (!m->d_wsm) {
            FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_wsm), nw * sizeof(float)));
            FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_wsm16), nw * sizeof(float)));
            m->d_wsm_numel = nw;
        }
        for (auto& r : pk.smallm) {
            if ((int64_t)r.Cout * r.taps * r.Cin * 4 >= (1LL << 31)) continue;
            int src = launch_smallm_weights(m->d_w + r.off, r.Cout, r.taps, r.Cin, m->d_wsm + r.off, nullptr);
            if (src == OK) src = launch_smallm_weights(m->d_w + r.off, r.Cout, r.taps, r.Cin, m->d_wsm16 + r.off, nullptr, true);
            if (src != OK) return src;
            m->wsm_offs.insert(r.off);
        }
    }
    // F(4,3) x F(2,3) weights of the same convs (the two-dimensional kernel takes the large images)
    m->wino2_of.clear();
    if (conv_w2d_enabled()) {
        int64_t w2_total = 0;
        for (auto& r : pk.wino) {
            if ((int64_t)r.Cout * 24 * r.Cin * 4 >= (1LL << 31)) continue;
            m->wino2_of[r.off] = w2_total;
            w2_total += (conv_w2d_numel(r.Cout, r.Cin) + 63) & ~(int64_t)63;
        }
        if (m->d_wino2 && m->d_wino2_numel < w2_total) {
            FLOWSE_HIP(hipFree(m->d_wino2));
            m->d_wino2 = nullptr;
        }
        if (!m->d_wino2 && w2_total > 0) {
            FLOWSE_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_wino2), w2_total * sizeof(float)));
            m->d_wino2_numel = w2_total;
        }
        for (auto& r : pk.wino) {
            const auto it = m->wino2_of.find(r.off);
            if (it == m->wino2_of.end()) continue;
            const int wrc = launch_w2d_weights(m->d_w + r.off, r.Cout, r.Cin, m->d_wino2 + it->second, nullptr);
            if (wrc != OK) return wrc;
        }
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
    if (conv_smallm_ok(B, H, W, Cin, 0, Cout, taps) || conv1x1_stream_ok(B, H, W, Cin, 0, Cout, taps))
        return (int64_t)Cout * taps * Cin;                                                  // fragment-order weight copy
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
    if (splitk_scratch && conv1x1_stream_ok(B, H, W, c.C1, c.C2, Cout, taps)) {   // the model handle's kernel for this shape
        const int rc = launch_smallm_weights(w, Cout, taps, c.C1 + c.C2, splitk_scratch, s, false);
        if (rc != OK) return rc;
        c.wsm = splitk_scratch;
    } else if (splitk_scratch && conv_smallm_ok(B, H, W, c.C1, c.C2, Cout, taps)) {
        const bool t16 = conv_smallm_tile16(B, H, W);
        const int rc = launch_smallm_weights(w, Cout, taps, c.C1 + c.C2, splitk_scratch, s, t16);
        if (rc != OK) return rc;
        c.wsm = splitk_scratch;
        c.wsm16 = t16 ? splitk_scratch : nullptr;
    } else if (splitk_scratch) {
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

static int op_conv3x3_winograd(const float* in1, int C1, const float* in2, int C2, const float* gamma,
                               const float* beta, float eps, int silu, const float* w, const float* bias,
                               const float* bias2, int bias2_stride, const float* res, float* out, int B, int H, int W,
                               int Cout, float scale, float* scratch, void* stream, bool two_d = false) {
    if (!in1 || !w || !out || !scratch || (gamma && !beta)) {
        set_error("flowse_op_conv3x3_f43: null argument");
        return ERR_ARG;
    }
    if (!in2) C2 = 0;
    if (two_d ? !conv_w2d_shape_ok(B, H, W, C1, C2, Cout, 9) : !conv_supports_wino(B, H, W, C1, C2, Cout, 9)) {
        set_error("flowse_op_conv3x3_%s: shape B=%d H=%d W=%d C=%d+%d Cout=%d not covered by the Winograd kernel",
                  two_d ? "w2d" : "f43", B, H, W, C1, C2, Cout);
        return ERR_SHAPE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = C1 + C2, HW = H * W;
    float* wf = scratch + flowse_op_group_norm_scratch_floats(B, HW, C);
    int rc = two_d ? launch_w2d_weights(w, Cout, C, wf, s) : launch_f43_weights(w, Cout, C, wf, s);
    if (rc != OK) return rc;
    ConvArgs c;
    c.in1 = in1; c.in2 = in2; c.C1 = C1; c.C2 = C2;
    c.w = w; c.bias = bias; c.bias2 = bias2; c.bias2_stride = bias2_stride; c.res = res; c.out = out;
    c.B = B; c.H = H; c.W = W; c.Cout = Cout; c.taps = 9; c.scale = scale;
    if (two_d) c.wino2 = wf;
    else c.wino = wf;
    const int ks = two_d ? 1 : conv_ksplit(B, H, W, C, Cout, 9);
    if (ks > 1) {                           // same split plan as the model handle uses for this shape
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

int64_t flowse_op_conv3x3_f43_scratch_floats(int B, int H, int W, int C, int Cout) {
    const int ks = conv_ksplit(B, H, W, C, Cout, 9);                   // > 1: F(4,3) runs split over K (small images)
    return flowse_op_group_norm_scratch_floats(B, H * W, C) + conv_wino_numel(Cout, C) +
           (ks > 1 ? (int64_t)ks * B * H * W * Cout : 0);
}

int64_t flowse_op_conv3x3_w2d_scratch_floats(int B, int H, int W, int C, int Cout) {
    return flowse_op_group_norm_scratch_floats(B, H * W, C) + conv_w2d_numel(Cout, C);
}

int flowse_op_conv3x3_w2d(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                          float eps, int silu, const float* w, const float* bias, const float* bias2, int bias2_stride,
                          const float* res, float* out, int B, int H, int W, int Cout, float scale, float* scratch,
                          void* stream) {
    return op_conv3x3_winograd(in1, C1, in2, C2, gamma, beta, eps, silu, w, bias, bias2, bias2_stride, res, out, B, H,
                               W, Cout, scale, scratch, stream, true);
}

int flowse_op_conv3x3_f43(const float* in1, int C1, const float* in2, int C2, const float* gamma, const float* beta,
                          float eps, int silu, const float* w, const float* bias, const float* bias2, int bias2_stride,
                          const float* res, float* out, int B, int H, int W, int Cout, float scale, float* scratch,
                          void* stream) {
    return op_conv3x3_winograd(in1, C1, in2, C2, gamma, beta, eps, silu, w, bias, bias2, bias2_stride, res, out, B, H,
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
    // the progressive-output heads (C -> 4, ncsnpp.py:345-366): 16-bit input, fp32 residual (the pyramid) and fp32 output, as
    // in the model -- conv3x3_head4_16_kernel
    const bool head = Cout == 4 && taps == 9 && !in2 && conv_supports_head4(B, H, W, C1, 0, Cout, taps);
    const int ks = (halo || head) ? 1 : conv16_ksplit(B, H, W, (int)C, Cout, taps);
    const int64_t nw = ((int64_t)Cout * taps * C + 3) & ~(int64_t)3;
    // the same 256-byte round-up per sub-buffer as take() below, upper bound over the optional ones
    auto up = [](int64_t b) { return (b + 255) & ~(int64_t)255; };
    const int64_t need = up(2 * M * C1) + up(2 * M * C2) + 2 * up(2 * nw) + 2 * up(2 * M * Cout) +
                         (ks > 1 ? up(4 * (int64_t)ks * M * Cout) : 0);
    if (scratch_bytes < need) {
        set_error("flowse_op_conv2d_16: scratch needs %lld bytes", (long long)need);
        return ERR_ARG;
    }
    if (gn_mean && !halo && !head) {
        set_error("flowse_op_conv2d_16: fused GroupNorm input only on LDS-halo shapes");
        return ERR_SHAPE;
    }
    char* p = static_cast<char*>(scratch);
    auto take = [&](int64_t bytes) { char* q = p; p += (bytes + 255) & ~(int64_t)255; return q; };
    void* a1 = take(2 * M * C1);
    void* a2 = C2 ? take(2 * M * C2) : nullptr;
    void* wq = take(2 * nw);
    const bool frag = (taps == 9 && conv16_uses_pc(B, H, W, C1, C2, Cout, taps)) || conv16_smallm_ok(B, H, W, C1, C2, Cout, taps);
    void* wfrag = frag ? take(2 * nw) : nullptr;
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
    if (rc == OK && frag) rc = launch_pc16_weights(wq, Cout, (int)C, wfrag, s, taps);
    if (rc == OK && res) rc = launch_convert(res, DT_F32, r16, dt, M * Cout, s);
    if (rc != OK) return rc;
    ConvArgs c;
    c.in1 = static_cast<const float*>(a1); c.in2 = static_cast<const float*>(a2); c.C1 = C1; c.C2 = C2;
    c.w = w; c.bias = bias; c.bias2 = nullptr; c.bias2_stride = 0;
    c.res = static_cast<const float*>(r16); c.out = static_cast<float*>(o16);
    c.B = B; c.H = H; c.W = W; c.Cout = Cout; c.taps = taps; c.scale = scale;
    c.ksplit = ks; c.partial = part;
    c.wq = wq; c.terms = 1; c.wq_f16 = dt == DT_F16 ? 1 : 0;
    c.wfrag = wfrag;
    c.in_dt = dt; c.out_dt = dt;
    if (gn_mean) {
        c.gn = GnParams{gn_mean, gn_scale, gn_beta};
        c.gn_silu = silu;
    }
    if (head) {
        c.res = res;
        c.out = out;
        c.out_dt = DT_F32;
    }
    rc = launch_conv(c, s);
    if (rc != OK || head) return rc;
    return launch_convert(o16, dt, out, DT_F32, M * Cout, s);
}

// Tail of ResnetBlockBigGANpp in 16-bit storage as ONE launch of the producer / consumer kernel:
//   out = (conv3x3(act(GroupNorm(h)); w1) + b1 + conv1x1(cat[x1, x2]; w2) + b2) * scale       (layerspp.py:265-274)
// with the shortcut as extra K steps (ConvArgs::sc1).  fp32 tensors at the boundary as in flowse_op_conv2d_16.
int flowse_op_resblock_tail_16(const float* h, int C, const float* gn_mean, const float* gn_scale, const float* gn_beta,
                               int silu, const float* w1, const float* b1, const float* x1, int XC1, const float* x2,
                               int XC2, const float* w2, const float* b2, float* out, int B, int H, int W, int Cout,
                               float scale, int dt, void* scratch, int64_t scratch_bytes, void* stream) {
    if (!h || !w1 || !x1 || !w2 || !out || !scratch || (dt != DT_BF16 && dt != DT_F16)) {
        set_error("flowse_op_resblock_tail_16: bad argument");
        return ERR_ARG;
    }
    if (!x2) XC2 = 0;
    if (!conv16_uses_pc(B, H, W, C, 0, Cout, 9)) {
        set_error("flowse_op_resblock_tail_16: only shapes conv3x3_pc16_kernel takes (H, W multiples of 16, C %% 32 == 0, "
                  "Cout %% 128 == 0, >= 64 tile x channel-block items)");
        return ERR_SHAPE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t M = (int64_t)B * H * W, XC = (int64_t)XC1 + XC2;
    const int64_t nw1 = ((int64_t)Cout * 9 * C + 3) & ~(int64_t)3, nw2 = ((int64_t)Cout * XC + 3) & ~(int64_t)3;
    auto up = [](int64_t b) { return (b + 255) & ~(int64_t)255; };
    const int64_t need = up(2 * M * C) + up(2 * M * XC1) + up(2 * M * XC2) + 2 * up(2 * nw1) + 2 * up(2 * nw2) + up(2 * M * Cout);
    if (scratch_bytes < need) {
        set_error("flowse_op_resblock_tail_16: scratch needs %lld bytes", (long long)need);
        return ERR_ARG;
    }
    char* p = static_cast<char*>(scratch);
    auto take = [&](int64_t bytes) { char* q = p; p += (bytes + 255) & ~(int64_t)255; return q; };
    void* h16 = take(2 * M * C);
    void* a1 = take(2 * M * XC1);
    void* a2 = XC2 ? take(2 * M * XC2) : nullptr;
    void* wq1 = take(2 * nw1);
    void* wf1 = take(2 * nw1);
    void* wq2 = take(2 * nw2);
    void* wf2 = take(2 * nw2);
    void* o16 = take(2 * M * Cout);
    int rc = launch_convert(h, DT_F32, h16, dt, M * C, s);
    if (rc == OK) rc = launch_convert(x1, DT_F32, a1, dt, M * XC1, s);
    if (rc == OK && XC2) rc = launch_convert(x2, DT_F32, a2, dt, M * XC2, s);
    if (rc == OK) rc = launch_convert(w1, DT_F32, wq1, dt, nw1, s);
    if (rc == OK) rc = launch_convert(w2, DT_F32, wq2, dt, nw2, s);
    if (rc == OK) rc = launch_pc16_weights(wq1, Cout, C, wf1, s, 9);
    if (rc == OK) rc = launch_pc16_weights(wq2, Cout, (int)XC, wf2, s, 1);
    if (rc != OK) return rc;
    ConvArgs c;
    c.in1 = static_cast<const float*>(h16); c.in2 = nullptr; c.C1 = C; c.C2 = 0;
    c.w = w1; c.bias = b1; c.bias_x = b2; c.bias2 = nullptr; c.bias2_stride = 0; c.res = nullptr;
    c.out = static_cast<float*>(o16);
    c.B = B; c.H = H; c.W = W; c.Cout = Cout; c.taps = 9; c.scale = scale;
    c.wq = wq1; c.terms = 1; c.wq_f16 = dt == DT_F16 ? 1 : 0;
    c.wfrag = wf1;
    c.sc1 = a1; c.SC1 = XC1; c.sc2 = a2; c.SC2 = XC2; c.wfrag_sc = wf2;
    c.in_dt = dt; c.out_dt = dt;
    if (gn_mean) {
        c.gn = GnParams{gn_mean, gn_scale, gn_beta};
        c.gn_silu = silu;
    }
    rc = launch_conv(c, s);
    if (rc != OK) return rc;
    return launch_convert(o16, dt, out, DT_F32, M * Cout, s);
}

int flowse_op_pc16_channel_blocks(int mode) {
    pc16_set_channel_blocks(mode);
    return OK;
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
