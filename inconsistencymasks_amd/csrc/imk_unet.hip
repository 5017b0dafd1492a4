// Host orchestration of the tiny U-Net (unet.py:4-67) on top of the kernels in imk_conv.hip /
// imk_elem.hip: topology, workspace layout, batched inference, ensemble inference + IM, and the training step
// (forward with batch statistics, loss, backward, AdamW).  The layer-level helpers are shared with EvalNet (imk_net.h).
// Everything is enqueued on the caller's stream; nothing here allocates or synchronises.
#include <atomic>
#include <cstdio>
#include "imk_net.h"
#include "imk_head.h"

namespace {

// ---- topology ------------------------------------------------------------------------------------
// index of every layer by role

struct Topo {
    int in_c, in_bn;
    int e_c3[4], e_c1[4], e_bn[4];
    int b_c3, b_c1, b_bn;
    int d_ca[4], d_bna[4], d_c3[4], d_c1[4], d_bnb[4];
    int out;
};

Topo make_topo(const imk_unet_plan *p) {
    Topo t{};
    t.in_c = p->find("in.c"); t.in_bn = p->find("in.bn");
    char nm[16];
    for (int i = 0; i < 4; ++i) {
        snprintf(nm, sizeof nm, "e%d.c3", i + 1); t.e_c3[i] = p->find(nm);
        snprintf(nm, sizeof nm, "e%d.c1", i + 1); t.e_c1[i] = p->find(nm);
        snprintf(nm, sizeof nm, "e%d.bn", i + 1); t.e_bn[i] = p->find(nm);
    }
    t.b_c3 = p->find("b.c3"); t.b_c1 = p->find("b.c1"); t.b_bn = p->find("b.bn");
    for (int j = 0; j < 4; ++j) {
        snprintf(nm, sizeof nm, "d%d.ca", j + 6); t.d_ca[j] = p->find(nm);
        snprintf(nm, sizeof nm, "d%d.bna", j + 6); t.d_bna[j] = p->find(nm);
        snprintf(nm, sizeof nm, "d%d.c3", j + 6); t.d_c3[j] = p->find(nm);
        snprintf(nm, sizeof nm, "d%d.c1", j + 6); t.d_c1[j] = p->find(nm);
        snprintf(nm, sizeof nm, "d%d.bnb", j + 6); t.d_bnb[j] = p->find(nm);
    }
    t.out = p->find("out");
    return t;
}

void build_layers(imk_unet_plan *p) {
    const int *ch = p->cfg.ch;  // 16a 32a 64a 128a 256a
    int c = add_conv(p, "in.c", 1, p->cfg.c_in, ch[0], 0);
    add_bn(p, "in.bn", ch[0], 0, c);
    const int enc_in[4] = {ch[0], ch[0], ch[1], ch[2]};
    const int enc_f[4] = {ch[0], ch[1], ch[2], ch[3]};
    char nm[16];
    for (int i = 0; i < 4; ++i) {
        snprintf(nm, sizeof nm, "e%d.c3", i + 1); add_conv(p, nm, 3, enc_in[i], enc_f[i], i);
        snprintf(nm, sizeof nm, "e%d.c1", i + 1); c = add_conv(p, nm, 1, enc_f[i], enc_f[i], i);
        snprintf(nm, sizeof nm, "e%d.bn", i + 1); add_bn(p, nm, enc_f[i], i, c);
    }
    add_conv(p, "b.c3", 3, ch[3], ch[4], 4);
    c = add_conv(p, "b.c1", 1, ch[4], ch[3], 4);
    add_bn(p, "b.bn", ch[3], 4, c);
    const int dec_in[4] = {ch[3], ch[2], ch[1], ch[0]};
    const int dec_f1[4] = {ch[3], ch[2], ch[1], ch[0]};
    const int dec_f2[4] = {ch[2], ch[1], ch[0], ch[0]};
    for (int j = 0; j < 4; ++j) {
        const int res = 3 - j;
        snprintf(nm, sizeof nm, "d%d.ca", j + 6); c = add_conv(p, nm, 1, dec_in[j], dec_f1[j], res);
        snprintf(nm, sizeof nm, "d%d.bna", j + 6); add_bn(p, nm, dec_f1[j], res, c);
        snprintf(nm, sizeof nm, "d%d.c3", j + 6); add_conv(p, nm, 3, dec_f1[j], dec_f1[j], res);
        snprintf(nm, sizeof nm, "d%d.c1", j + 6); c = add_conv(p, nm, 1, dec_f1[j], dec_f2[j], res);
        snprintf(nm, sizeof nm, "d%d.bnb", j + 6); add_bn(p, nm, dec_f2[j], res, c);
    }
    c = add_conv(p, "out", 1, ch[0], p->cfg.n_out, 0);
    p->layers[c].flags = IMK_LF_HEAD;

    // where every conv reads its input (the producer's BatchNorm, pooling and upsample+add are applied on load)
    const Topo t = make_topo(p);
    set_src(p, t.in_c, LM_U8, IMK_SRC_XA);
    set_src(p, t.e_c3[0], LM_AFFINE, t.in_c, t.in_bn);
    for (int i = 1; i < 4; ++i) set_src(p, t.e_c3[i], LM_POOL, t.e_c1[i - 1], t.e_bn[i - 1]);
    for (int i = 0; i < 4; ++i) set_src(p, t.e_c1[i], LM_RAW, t.e_c3[i]);
    set_src(p, t.b_c3, LM_POOL, t.e_c1[3], t.e_bn[3]);
    set_src(p, t.b_c1, LM_RAW, t.b_c3);
    for (int j = 0; j < 4; ++j) {
        const int lo_c = j == 0 ? t.b_c1 : t.d_c1[j - 1], lo_bn = j == 0 ? t.b_bn : t.d_bnb[j - 1];
        set_src(p, t.d_ca[j], LM_UPADD, lo_c, lo_bn, t.e_c1[3 - j], t.e_bn[3 - j]);
        set_src(p, t.d_c3[j], LM_AFFINE, t.d_ca[j], t.d_bna[j]);
        set_src(p, t.d_c1[j], LM_RAW, t.d_c3[j]);
    }
    set_src(p, t.out, LM_AFFINE, t.d_c1[3], t.d_bnb[3]);
    finish_layout(p);
}

Ws make_ws(const imk_unet_plan *p, int B, int mode) {
    Ws w;
    const Topo t = make_topo(p);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up(off + bytes); return o; };
    make_ws_layers(p, B, mode, w, take);
    if (mode == 0) { w.sched = take(IMK_SCHED_BYTES * p->layers.size()); w.has_sched = true; }

    if (mode == 1) {
        for (int j = 0; j < 4; ++j) {  // decoder j+6 at res 3-j; u has the channels of ca's input
            const ImkLayer &ca = p->layers[t.d_ca[j]];
            const Dim d = res_dim(p->cfg, ca.res);
            w.dU[j] = take((size_t)B * d.h * d.w * imk_pad8(ca.cin) * 2);
        }
        for (int i = 0; i < 4; ++i) {  // pooled output of encoder i+1, at res i+1
            const ImkLayer &e = p->layers[t.e_c1[i]];
            const Dim d = res_dim(p->cfg, i + 1);
            w.dP[i] = take((size_t)B * d.h * d.w * imk_pad8(e.cout) * 2);
        }
        const size_t px = (size_t)B * p->cfg.h * p->cfg.w;
        w.dlogit = take(px * imk_pad8(p->cfg.n_out) * 2);
        w.loss_partial = take((size_t)imk_loss_blocks((long long)px) * sizeof(float));
        w.probs = take(px * p->cfg.n_out * sizeof(float));
    }
    w.total = off;
    return w;
}

int run_forward(Ctx &c, const Topo &t, float *probs, float *params_rw) {
    int rc;
    // inference: the tile counters of this forward's launches (dynamic walk of the persistent conv kernels) start at zero
    if (!c.train && c.ws.has_sched) IMK_HIP(hipMemsetAsync(c.base + c.ws.sched, 0, IMK_SCHED_BYTES * c.p->layers.size(), c.stream));
#define RUN(conv) do { rc = run_conv_fwd(c, (conv), params_rw); if (rc) return rc; } while (0)
    // Conv3x3+ReLU -> Conv1x1+ReLU pairs run as one kernel where the channel counts allow it
#define RUN_PAIR(c3, c1) do { rc = run_conv_pair(c, (c3), (c1), params_rw); if (rc) return rc; } while (0)
    // Inference: the input block (x/255 -> Conv1x1+ReLU -> BN, unet.py:4-9) is computed by the first encoder conv while it
    // stages its tile (LM_STEM), so the full-resolution stem tensor is neither written nor read.
    bool stem_fused = false;
    {
        const ImkLayer &st = c.p->layers[t.in_c], &c3 = c.p->layers[t.e_c3[0]];
        if (!c.train && !c.p->dbg_materialize && imk_conv_stem_fusable(st.cin, st.cout, c3.cout)) {
            ImkInput x{};
            x.in = c.x_in[0]; x.lmode = LM_STEM; x.cin = c3.cin; x.cs_in = imk_pad8(c3.cin); x.u8_c = st.cin;
            x.sc = c.bn_scale(t.in_bn); x.sh = c.bn_shift(t.in_bn);
            x.sc2 = c.params + st.off_w; x.sh2 = c.params + st.off_b;
            bool f = false;
            rc = run_conv_fwd(c, t.e_c3[0], params_rw, t.e_c1[0], &f, &x);
            if (rc == IMK_OK) stem_fused = true;
            else if (rc != IMK_EUNSUPPORTED) return rc;
        }
    }
    if (!stem_fused) { RUN(t.in_c); RUN_PAIR(t.e_c3[0], t.e_c1[0]); }
    for (int i = 1; i < 4; ++i) RUN_PAIR(t.e_c3[i], t.e_c1[i]);
    RUN_PAIR(t.b_c3, t.b_c1);
    for (int j = 0; j < 4; ++j) {
        rc = run_conv_pre_pair(c, t.d_ca[j], t.d_c3[j], t.d_c1[j]);      // inference, shallow levels: one launch per decoder block
        if (rc == IMK_OK) continue;
        if (rc != IMK_EUNSUPPORTED) return rc;
        RUN(t.d_ca[j]); RUN_PAIR(t.d_c3[j], t.d_c1[j]);
    }
#undef RUN_PAIR
#undef RUN
    if (!probs) return IMK_OK;   // training: the caller runs the fused head + loss kernel
    const ImkLayer &o = c.p->layers[t.out];
    const int bn = t.d_bnb[3];
    return imk_launch_head(c.act(t.d_c1[3]), c.bn_scale(bn), c.bn_shift(bn), c.params + o.off_w, c.params + o.off_b,
                           o.cin, imk_pad8(o.cin), o.cout, c.p->cfg.act_out, (long long)c.B * c.p->cfg.h * c.p->cfg.w,
                           probs, c.stream);
}

bool cfg_ok(const imk_unet_cfg *c) {
    if (!c) return false;
    if (c->h <= 0 || c->w <= 0 || (c->h % 16) || (c->w % 16)) return false;
    if (c->c_in < 1 || c->c_in > 8 || c->n_out < 1 || c->n_out > 64) return false;
    for (int i = 0; i < 5; ++i) if (c->ch[i] < 1 || c->ch[i] > 512) return false;
    if (imk_pad8(c->ch[0]) > 32) return false;  // head kernel instantiations
    return c->act_out == 0 || c->act_out == 1;
}

}  // namespace

// =====================================================================================================
// The process-wide side-stream pool (see imk_net.h: why one pool and not one pair of streams per plan).
static std::atomic<int> g_runtime_warnings{0};
hipStream_t imk_side_pool_stream(int i) {
    struct Pool { std::mutex mu; hipStream_t s[imk_unet_plan::MAX_SIDE] = {}; };
    static Pool pools[64];
    if (i < 0 || i >= imk_unet_plan::MAX_SIDE) return nullptr;
    // The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read when it starts.  With fewer than 8, a
    // side stream can land on the queue of the stream it is meant to run beside and the "concurrent" weight gradients queue up
    // behind barrier packets (-14 % on the ISIC generation with an RCCL stream in the process).  The Python package
    // exports 8 before the runtime starts; a bare C-ABI caller is told here, once.
    static std::once_flag hwq_once;
    std::call_once(hwq_once, []() {
        const char *e = getenv("GPU_MAX_HW_QUEUES");
        if (!e || atoi(e) < 8) {
            g_runtime_warnings.fetch_or(IMK_WARN_HW_QUEUES);
            fprintf(stderr, "libimk: GPU_MAX_HW_QUEUES=%s (< 8): side streams may serialise behind the caller's stream; "
                            "export GPU_MAX_HW_QUEUES=8 before the HIP runtime starts\n", e ? e : "unset");
        }
    });
    int dev = 0;
    (void)hipGetDevice(&dev);
    Pool &pool = pools[dev & 63];
    std::lock_guard<std::mutex> lock(pool.mu);
    if (!pool.s[i] && hipStreamCreateWithFlags(&pool.s[i], hipStreamNonBlocking) != hipSuccess) pool.s[i] = nullptr;
    return pool.s[i];
}

extern "C" int imk_runtime_warnings(void) { return g_runtime_warnings.load(); }

extern "C" int imk_unet_plan_side_stream(const imk_unet_plan *plan, int i, void **stream_out) {
    IMK_CHECK_ARG(plan && stream_out && i >= 0 && i < imk_unet_plan::MAX_SIDE);
    if (!ensure_side_streams(plan, i + 1)) return IMK_EINVAL;
    *stream_out = (void *)plan->side[i];
    return IMK_OK;
}

extern "C" int imk_unet_plan_create(const imk_unet_cfg *cfg, imk_unet_plan **out) {
    IMK_CHECK_ARG(out);
    if (!cfg_ok(cfg)) return IMK_EINVAL;
    if ((long long)cfg->h * cfg->w >= imk_conv_max_pixels() || cfg->w >= (1 << 16)) return IMK_EUNSUPPORTED;   // include/imk.h: image size limit
    imk_unet_plan *p = new (std::nothrow) imk_unet_plan();
    if (!p) return IMK_EINVAL;
    p->cfg = *cfg;
    build_layers(p);
    *out = p;
    return IMK_OK;
}

extern "C" void imk_unet_plan_destroy(imk_unet_plan *plan) {
    if (!plan) return;
    for (int i = 0; i < imk_unet_plan::MAX_SIDE; ++i) {
        if (!plan->side[i]) continue;
        (void)hipStreamSynchronize(plan->side[i]);      // the streams belong to the process-wide pool (imk_net.h)
        if (plan->ev_join[i]) (void)hipEventDestroy(plan->ev_join[i]);
    }
    for (auto &e : plan->ev_fork) if (e) (void)hipEventDestroy(e);
    for (auto &e : plan->ev_ring) if (e) (void)hipEventDestroy(e);
    delete plan;
}

extern "C" int imk_unet_param_count(const imk_unet_plan *plan, int64_t *total, int64_t *trainable) {
    IMK_CHECK_ARG(plan);
    if (total) *total = plan->n_total;
    if (trainable) *trainable = plan->n_trainable;
    return IMK_OK;
}

extern "C" int imk_unet_num_layers(const imk_unet_plan *plan) { return plan ? (int)plan->layers.size() : IMK_EINVAL; }

extern "C" int imk_unet_layer_info(const imk_unet_plan *plan, int idx, imk_layer_info *out) {
    IMK_CHECK_ARG(plan && out && idx >= 0 && idx < (int)plan->layers.size());
    const ImkLayer &l = plan->layers[idx];
    memset(out, 0, sizeof *out);
    strncpy(out->name, l.name.c_str(), sizeof(out->name) - 1);
    out->kind = l.kind; out->ksize = l.ksize; out->cin = l.cin; out->cout = l.cout;
    out->off_w = l.off_w; out->off_b = l.off_b; out->off_mean = l.off_mean; out->off_var = l.off_var;
    return IMK_OK;
}

extern "C" int imk_unet_plan_set_bn_momentum(imk_unet_plan *plan, float momentum) {
    IMK_CHECK_ARG(plan && momentum >= 0.f && momentum < 1.f);
    plan->bn_momentum = momentum;
    return IMK_OK;
}

extern "C" int imk_unet_plan_get_bn_momentum(const imk_unet_plan *plan, float *momentum) {
    IMK_CHECK_ARG(plan && momentum);
    *momentum = plan->bn_momentum;
    return IMK_OK;
}

extern "C" int imk_unet_plan_debug(imk_unet_plan *plan, int materialize, int single_stream) {
    IMK_CHECK_ARG(plan);
    if (materialize >= 0) plan->dbg_materialize = materialize != 0;
    if (single_stream >= 0) plan->dbg_single_stream = single_stream != 0;
    return IMK_OK;
}

extern "C" int64_t imk_unet_packed_bytes(const imk_unet_plan *plan) { return plan ? plan->packed_bytes : IMK_EINVAL; }


extern "C" int imk_unet_pack_weights(const imk_unet_plan *plan, const float *params, void *packed, void *stream_) {
    return pack_weights(plan, params, packed, (hipStream_t)stream_, nullptr, nullptr, true);
}

extern "C" int64_t imk_unet_workspace_bytes(const imk_unet_plan *plan, int batch, int mode) {
    if (!plan || batch <= 0 || (mode != 0 && mode != 1)) return IMK_EINVAL;
    return (int64_t)make_ws(plan, batch, mode).total;
}

extern "C" int imk_unet_forward(const imk_unet_plan *plan, const float *params, const void *packed, const uint8_t *x,
                                int batch, float *probs, void *workspace, int64_t workspace_bytes, void *stream_) {
    IMK_CHECK_ARG(plan && params && packed && x && probs && workspace && batch > 0);
    Ctx c{plan, make_ws(plan, batch, 0), (uint8_t *)workspace, params, (const uint8_t *)packed, batch, false,
          (hipStream_t)stream_};
    if ((int64_t)c.ws.total > workspace_bytes) return IMK_EWORKSPACE;
    c.x_in[0] = x;
    return run_forward(c, make_topo(plan), probs, nullptr);
}

extern "C" int imk_unet_tensor_info(const imk_unet_plan *plan, int batch, int mode, int layer_idx, int which,
                                    int64_t *byte_offset, int *h, int *w, int *c, int *c_stride) {
    IMK_CHECK_ARG(plan && batch > 0 && layer_idx >= 0 && layer_idx < (int)plan->layers.size());
    IMK_CHECK_ARG(mode == 0 || mode == 1);
    const Ws ws = make_ws(plan, batch, mode);
    const ImkLayer &l = plan->layers[layer_idx];
    const Dim d = res_dim(plan->cfg, l.res);
    size_t off = 0;
    if (which == 0 && l.kind == 0 && layer_idx != plan->find("out")) off = ws.L[layer_idx].out;
    else if (which == 1 && l.kind == 0 && mode == 1) off = ws.L[layer_idx].dA;
    else if (which == 2 && l.kind == 1 && mode == 1) off = ws.L[layer_idx].dy;
    else return IMK_EINVAL;
    if (byte_offset) *byte_offset = (int64_t)off;
    if (h) *h = d.h;
    if (w) *w = d.w;
    if (c) *c = l.cout;
    if (c_stride) *c_stride = imk_pad8(l.cout);
    return IMK_OK;
}

// Ensemble inference + IM.  Workspace: n_models head slabs, then one activation workspace per concurrent stream.  A head
// slab holds what the IM stage reads of one model: its last decoder activation [B,H,W,cs] fp16 (fused head + IM kernel,
// the default) or its fp32 probabilities [B,H,W,K] (unfused: head_kernel per model + imk_im_*; taken for shapes the fused
// kernel does not cover, under imk_debug_materialize(1), and when the caller sized the slabs for probabilities only).
static size_t head_slab_bytes(const imk_unet_plan *plan, int batch, bool fused) {
    const imk_unet_cfg &cf = plan->cfg;
    const size_t px = (size_t)batch * cf.h * cf.w;
    const size_t probs = px * cf.n_out * sizeof(float), zlast = px * imk_pad8(cf.ch[0]) * sizeof(f16);
    return up(fused && zlast > probs ? zlast : probs);
}

extern "C" int64_t imk_unet_forward_im_workspace_bytes(const imk_unet_plan *plan, int n_models, int batch, int n_streams) {
    if (!plan || plan->net != 0 || n_models <= 0 || batch <= 0 || n_streams <= 0) return IMK_EINVAL;
    return (int64_t)(head_slab_bytes(plan, batch, true) * n_models + make_ws(plan, batch, 0).total * n_streams);
}

extern "C" int imk_unet_forward_im(const imk_unet_plan *plan, int n_models, const float *const *params,
                                   const void *const *packed, const uint8_t *x, int batch, float thr, int cmp_ge,
                                   const uint8_t *img, int block_in, int block_out, uint8_t *img_out, uint8_t *masks_out,
                                   uint8_t *im_out, int64_t *im_size, int64_t *pred_size, uint8_t *presence,
                                   void *workspace, int64_t workspace_bytes, void *stream_) {
    IMK_CHECK_ARG(plan && params && packed && x && workspace && batch > 0 && n_models > 0);
    const imk_unet_cfg &cf = plan->cfg;
    const Ws ws = make_ws(plan, batch, 0);
    const Topo topo = make_topo(plan);
    // fused head + IM?  (the kernel's shape limits, and room for the activation slabs)
    ImkHeadImArgs ha{};
    ha.n_models = n_models; ha.cin = cf.ch[0]; ha.cs = imk_pad8(cf.ch[0]); ha.K = cf.n_out; ha.softmax = cf.act_out;
    ha.batch = batch; ha.hw = cf.h * cf.w; ha.thr = thr; ha.cmp_ge = cmp_ge; ha.img = img; ha.c = cf.c_in;
    ha.block_in = block_in; ha.block_out = block_out; ha.img_out = img_out; ha.masks_out = masks_out; ha.im_out = im_out;
    ha.im_size = im_size; ha.pred_size = pred_size; ha.presence = presence;
    static const bool fuse_off = []() { const char *e = getenv("IMK_HEAD_IM_FUSE"); return e && e[0] == '0'; }();
    bool fused = !fuse_off && !plan->dbg_materialize && imk_head_im_supported(ha) &&
                 (int64_t)(head_slab_bytes(plan, batch, true) * n_models + ws.total) <= workspace_bytes;
    const size_t slab = head_slab_bytes(plan, batch, fused);
    if ((int64_t)(slab * n_models + ws.total) > workspace_bytes) return IMK_EWORKSPACE;
    // the unfused probability stack must be contiguous [N,B,H,W,K]: only the last slab may be padded
    const size_t probs_exact = (size_t)batch * cf.h * cf.w * cf.n_out * sizeof(float);
    uint8_t *base = (uint8_t *)workspace;
    // The models are independent until the IM stage: with room for one activation workspace per stream (the caller
    // passes slabs + k * imk_unet_workspace_bytes, k <= 1 + MAX_SIDE) they run on k streams side by side -- the deep
    // layers of one model fill the gaps of the other's.  With room for one only, they run back to back.
    const int max_streams = 1 + imk_unet_plan::MAX_SIDE;
    int n_slabs = (int)(((size_t)workspace_bytes - slab * n_models) / ws.total);
    if (n_slabs > n_models) n_slabs = n_models;
    if (n_slabs > max_streams) n_slabs = max_streams;
    static const bool conc_off = []() { const char *e = getenv("IMK_ENSEMBLE_STREAMS"); return e && e[0] == '0'; }();
    if (n_slabs > 1 && (conc_off || plan->dbg_single_stream || !ensure_side_streams(plan, n_slabs - 1))) n_slabs = 1;
    hipStream_t main_stream = (hipStream_t)stream_;
    if (n_slabs > 1) {
        IMK_HIP(hipEventRecord(plan->ev_fork[0], main_stream));
        for (int s = 1; s < n_slabs; ++s) IMK_HIP(hipStreamWaitEvent(plan->side[s - 1], plan->ev_fork[0], 0));
    }
    const ImkLayer &o = plan->layers[topo.out];
    for (int m = 0; m < n_models; ++m) {
        const int sl = m % n_slabs;
        Ctx c{plan, ws, base + slab * n_models + ws.total * sl, params[m], (const uint8_t *)packed[m], batch, false,
              sl == 0 ? main_stream : plan->side[sl - 1]};
        c.x_in[0] = x;
        if (fused) {    // the last decoder activation goes to the model's head slab; no head launch
            c.ovr_conv = topo.d_c1[3];
            c.ovr_out = reinterpret_cast<f16 *>(base + slab * m);
            ha.z[m] = c.ovr_out;
            ha.sc[m] = c.bn_scale(topo.d_bnb[3]); ha.sh[m] = c.bn_shift(topo.d_bnb[3]);
            ha.w[m] = params[m] + o.off_w; ha.bias[m] = params[m] + o.off_b;
        }
        int rc = run_forward(c, topo, fused ? nullptr : (float *)(base + probs_exact * m), nullptr);
        if (rc) return rc;
    }
    for (int s = 1; s < n_slabs; ++s) {
        IMK_HIP(hipEventRecord(plan->ev_join[s - 1], plan->side[s - 1]));
        IMK_HIP(hipStreamWaitEvent(main_stream, plan->ev_join[s - 1], 0));
    }
    if (fused) return imk_launch_head_im(ha, main_stream);
    if (cf.act_out == 0)
        return imk_im_binary((const float *)base, n_models, batch, cf.h, cf.w, cf.n_out, thr, cmp_ge, img, cf.c_in, block_in,
                             block_out, img_out, masks_out, im_out, im_size, pred_size, stream_);
    return imk_im_multiclass((const float *)base, n_models, batch, cf.h, cf.w, cf.n_out, img, cf.c_in, block_in, block_out,
                             img_out, masks_out, im_out, im_size, presence, stream_);
}

// =====================================================================================================
// training
// =====================================================================================================

extern "C" int64_t imk_unet_state_bytes(const imk_unet_plan *plan) {
    if (!plan) return IMK_EINVAL;
    return (int64_t)(2 * up((size_t)plan->n_trainable * sizeof(float)) + up(sizeof(ImkCtl)));
}

extern "C" int imk_unet_state_init(const imk_unet_plan *plan, void *state, void *stream_) {
    IMK_CHECK_ARG(plan && state);
    hipStream_t stream = (hipStream_t)stream_;
    IMK_HIP(hipMemsetAsync(state, 0, (size_t)imk_unet_state_bytes(plan), stream));
    return imk_launch_ctl_init(state_view(plan, state).ctl, stream);
}

extern "C" int imk_unet_fwd_bwd(const imk_unet_plan *plan, float *params, void *packed, void *state, const uint8_t *x,
                                const uint8_t *y, int batch, int loss_kind, float *grads, float *stats, void *workspace,
                                int64_t workspace_bytes, void *stream_) {
    IMK_CHECK_ARG(plan && params && packed && state && x && y && grads && stats && workspace && batch > 0);
    IMK_CHECK_ARG(loss_kind == 0 || loss_kind == 1);
    IMK_CHECK_ARG((loss_kind == 0) == (plan->cfg.act_out == 0));  // mse <-> sigmoid head, cce <-> softmax head
    hipStream_t stream = (hipStream_t)stream_;
    Ctx c{plan, make_ws(plan, batch, 1), (uint8_t *)workspace, params, (const uint8_t *)packed, batch, true, stream};
    if ((int64_t)c.ws.total > workspace_bytes) return IMK_EWORKSPACE;
    c.x_in[0] = x;
    const StateView sv = state_view(plan, state);
    const imk_unet_cfg &cf = plan->cfg;
    const Topo t = make_topo(plan);
    const long long n_pix = (long long)batch * cf.h * cf.w;
    int rc;
#define OK(e) do { rc = (e); if (rc) return rc; } while (0)
    OK(run_forward(c, t, nullptr, params));

    // head + loss + d(loss*scale)/d(logits) in one pass over the last activation
    f16 *dlogit = reinterpret_cast<f16 *>(c.base + c.ws.dlogit);
    float *loss_partial = reinterpret_cast<float *>(c.base + c.ws.loss_partial);
    const ImkLayer &ol = plan->layers[t.out];
    const int obn = t.d_bnb[3];
    // softmax heads: the loss, the gradient of the last BatchNorm's output (with its statistics) and the output layer's weight
    // gradient come out of ONE kernel that reads the last activation once (imk_headf.hip); sigmoid heads: head_loss_kernel,
    // then the output layer's dgrad (+ fused weight gradient where the pipelined kernel covers the shape)
    const bool head_one_pass = cf.act_out == 1 &&
                               imk_head_cce_fused_ok(imk_pad8(ol.cin), ol.cout, n_pix, c.ws.L[obn].n_bwd_rows);
    // sigmoid heads with 1 / 3 maps on <= 16 channels (ISIC, HeLa): the same in one pass (head_mse_fused_kernel, round 4)
    const bool head_mse_pass = cf.act_out == 0 && imk_head_mse_fused_ok(imk_pad8(ol.cin), ol.cout, n_pix, c.ws.L[obn].n_bwd_rows);
    if (!head_one_pass && !head_mse_pass)
        OK(imk_launch_head_loss(c.act(t.d_c1[3]), c.bn_scale(obn), c.bn_shift(obn), params + ol.off_w, params + ol.off_b, ol.cin,
                                imk_pad8(ol.cin), ol.cout, cf.act_out, n_pix, y, sv.ctl, stats, dlogit, loss_partial, stream));

    // The weight-gradient kernels run on a side stream, forked after the kernel that produced their gradient operand and
    // joined before the final reduction: nothing on the backward chain depends on them, and they fill the gaps that the
    // latency-bound kernels of the chain leave (1.280 vs 1.365 ms per step with all 24 on the side stream; forking only
    // the deep layers' was neutral).  IMK_SIDE_PIXELS = largest B*H*W that is forked (0: single stream).
    static const long long side_px = []() { const char *e = getenv("IMK_SIDE_PIXELS"); return e ? atoll(e) : (1LL << 62); }();
    static const int n_side_env = []() { const char *e = getenv("IMK_SIDE_STREAMS"); int v = e ? atoi(e) : 1;
                                         return v < 0 ? 0 : (v > imk_unet_plan::MAX_SIDE ? imk_unet_plan::MAX_SIDE : v); }();
    const bool side_on = side_px > 0 && !plan->dbg_single_stream && n_side_env > 0 && ensure_side_streams(plan, n_side_env);
    Bwd b{c, grads, sv.ctl, stats + 1, side_on ? n_side_env : 0, side_px};
    ImkStopRingScope stop_ring(plan, stream, b.n_side);      // from here on: the backward pass's launches carry their own events
    bool loss_done = false;
    auto loss_on_side = [&]() -> int {      // once, as soon as a fork exists (every fork event is younger than the head's kernel)
        if (loss_done || b.n_side <= 0 || b.n_fork <= 0) return IMK_OK;
        loss_done = true;
        return imk_launch_loss_finalize(loss_partial, n_pix, cf.n_out, loss_kind, stats, plan->side[(b.n_fork - 1) % b.n_side]);
    };
    if (head_mse_pass) {
        const int rows = imk_head_cce_fused_rows(n_pix);
        float *wgp = reinterpret_cast<float *>(c.base + c.ws.L[t.out].wg_partial);
        OK(imk_launch_head_mse_fused(c.act(t.d_c1[3]), c.bn_scale(obn), c.bn_shift(obn), params + ol.off_w, params + ol.off_b,
                                     ol.cin, imk_pad8(ol.cin), ol.cout, n_pix, y, sv.ctl, stats, c.dy(obn), loss_partial,
                                     reinterpret_cast<float *>(c.base + c.ws.L[obn].bwd_partial), wgp, stream));
        b.dy_rows[obn] = rows;
        OK(imk_wgf_add_job(b.jobs, wgp, rows, ol.ksize, ol.cin, ol.cout, grads + ol.off_w, grads + ol.off_b));
    } else if (head_one_pass) {
        const int rows = imk_head_cce_fused_rows(n_pix);
        float *wgp = reinterpret_cast<float *>(c.base + c.ws.L[t.out].wg_partial);
        OK(imk_launch_head_cce_fused(c.act(t.d_c1[3]), c.bn_scale(obn), c.bn_shift(obn), params + ol.off_w, params + ol.off_b,
                                     ol.cin, imk_pad8(ol.cin), ol.cout, n_pix, y, sv.ctl, stats, c.dy(obn), loss_partial,
                                     reinterpret_cast<float *>(c.base + c.ws.L[obn].bwd_partial), wgp, stream));
        b.dy_rows[obn] = rows;
        OK(imk_wgf_add_job(b.jobs, wgp, rows, ol.ksize, ol.cin, ol.cout, grads + ol.off_w, grads + ol.off_b));
    } else {
        // head: its "dA" is dlogit.  Its dgrad (dlogit -> dy of the last BatchNorm, with that BN's gradient statistics) reads
        // exactly the operands of its weight gradient (dlogit, and z = the last decoder activation, whose BatchNorm output is
        // the head's input): where the pipelined kernel covers the shape, one launch produces both.
        ImkConvArgs ha{};
        ha.x.in = dlogit; ha.x.lmode = LM_RAW; ha.x.cin = ol.cout; ha.x.cs_in = imk_pad8(ol.cout);
        ha.B = batch; ha.H = cf.h; ha.W = cf.w; ha.ksize = 1; ha.cout = ol.cin; ha.cs_out = imk_pad8(ol.cin);
        ha.wpk = c.wbwd(t.out); ha.out = c.dy(obn); ha.epi = EP_PLAIN;
        ha.dystat_z = c.act(t.d_c1[3]);
        ha.stats_partial = reinterpret_cast<float *>(c.base + c.ws.L[obn].bwd_partial);
        const bool head_fused = imk_conv_can_fuse_wgrad(ha);
        if (!head_fused) OK(b.wgrad(t.out, dlogit));
        // the loss value only needs head_loss_kernel's partials: its reduction rides on the side stream, behind the head's
        // weight gradient, instead of sitting at the end of the step
        OK(loss_on_side());
        int rows = 0;
        ha.stats_rows = &rows;
        if (head_fused) {
            ha.wg_partial = reinterpret_cast<float *>(c.base + c.ws.L[t.out].wg_partial);
            ha.wg_sc = c.bn_scale(obn); ha.wg_sh = c.bn_shift(obn);
        }
        OK(imk_launch_conv(ha, stream));
        if (rows <= 0 || rows > c.ws.L[obn].n_bwd_rows) return IMK_EWORKSPACE;
        b.dy_rows[obn] = rows;
        if (head_fused) {
            if (rows > imk_conv_fused_wgrad_rows_max()) return IMK_EWORKSPACE;
            OK(imk_wgf_add_job(b.jobs, ha.wg_partial, rows, ol.ksize, ol.cin, ol.cout, grads + ol.off_w, grads + ol.off_b));
        }
    }
    // decoders 9..6
    for (int j = 3; j >= 0; --j) {
        f16 *dU = reinterpret_cast<f16 *>(c.base + c.ws.dU[j]);
        if (j == 3) OK(b.bn_bwd(t.d_bnb[j], 0, nullptr, nullptr));
        else OK(b.bn_bwd(t.d_bnb[j], 2, nullptr, reinterpret_cast<f16 *>(c.base + c.ws.dU[j + 1])));
        OK(b.wgrad_dgrad(t.d_c1[j], c.dA(t.d_c3[j]), c.act(t.d_c3[j])));
        OK(b.wgrad_dgrad(t.d_c3[j], c.dy(t.d_bna[j]), nullptr, t.d_bna[j]));
        OK(b.bn_bwd(t.d_bna[j], 0, nullptr, nullptr));
        b.sum2_bn = j > 0 ? t.d_bnb[j - 1] : t.b_bn;     // dU's 2x2 sums = dy of the block below: assembled by this dgrad where it can
        OK(b.wgrad_dgrad(t.d_ca[j], dU, nullptr));
        OK(b.flush_block());    // the block's weight gradients (and the head's, for the first block) -> side stream
        OK(loss_on_side());
    }
    // bottleneck: dy = 2x2 sum of dU[decoder 6]
    OK(b.bn_bwd(t.b_bn, 2, nullptr, reinterpret_cast<f16 *>(c.base + c.ws.dU[0])));
    OK(b.wgrad_dgrad(t.b_c1, c.dA(t.b_c3), c.act(t.b_c3)));
    OK(b.wgrad_dgrad(t.b_c3, reinterpret_cast<f16 *>(c.base + c.ws.dP[3]), nullptr));
    OK(b.flush_block());
    // encoders 4..1: dy = skip gradient (dU of decoder 6+(3-i)) + max-pool scatter of dP[i]
    for (int i = 3; i >= 0; --i) {
        if (i == 0) {
            b.defer_finalize = true;   // last block: nothing left to hide its split reductions behind
            static const bool swap_off = []() { const char *e = getenv("IMK_TAIL_SWAP"); return e && e[0] == '0'; }();
            if (!swap_off) b.hold_conv = t.e_c3[0];
        }
        OK(b.bn_bwd(t.e_bn[i], 1, reinterpret_cast<f16 *>(c.base + c.ws.dU[3 - i]),
                    reinterpret_cast<f16 *>(c.base + c.ws.dP[i])));
        OK(b.wgrad_dgrad(t.e_c1[i], c.dA(t.e_c3[i]), c.act(t.e_c3[i])));
        f16 *dst = i > 0 ? reinterpret_cast<f16 *>(c.base + c.ws.dP[i - 1]) : c.dy(t.in_bn);
        OK(b.wgrad_dgrad(t.e_c3[i], dst, nullptr, i > 0 ? -1 : t.in_bn));
        OK(i > 0 ? b.flush_block() : b.flush_wgrads());
    }
    OK(b.bn_bwd(t.in_bn, 0, nullptr, nullptr));
    if (b.has_held) {
        OK(b.wgrad(t.in_c));                        // full resolution: forked onto the side stream at once
        b.pending[b.n_pending++] = b.held;          // launched on the main stream by finish_wgrads
    } else {
        OK(b.wgrad(t.in_c, nullptr, true));
    }
    OK(b.finish_wgrads());
    if (!loss_done) OK(imk_launch_loss_finalize(loss_partial, n_pix, cf.n_out, loss_kind, stats, stream));
#undef OK
    return IMK_OK;
}

extern "C" int imk_unet_adamw_step(const imk_unet_plan *plan, float *params, void *packed, void *state,
                                   const float *grads, const float *stats, float grad_scale, float lr, float wd,
                                   float beta1, float beta2, float eps, void *stream_) {
    IMK_CHECK_ARG(plan && params && packed && state && grads && stats);
    const StateView sv = state_view(plan, state);
    int rc = imk_launch_adamw(params, sv.m, sv.v, grads, plan->n_trainable, sv.ctl, stats, grad_scale, lr, wd, beta1, beta2,
                              eps, (hipStream_t)stream_);
    if (rc) return rc;
    // conv weights only: training does not read the folded inference statistics (imk_unet_pack_weights refreshes them)
    return pack_weights(plan, params, packed, (hipStream_t)stream_, sv.ctl, stats, false);
}
