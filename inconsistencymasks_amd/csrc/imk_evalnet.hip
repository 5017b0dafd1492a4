// Host orchestration of EvalNet (evalnet.py:24-73: get_evalnet, get_evalnet_miou) on top of the same conv / BatchNorm
// kernels as the U-Net: two towers (input block + one conv block each), concatenate, five conv blocks, GlobalAvgPool2D,
// Dense sigmoid head(s).  A conv block is Conv3x3+ReLU -> Conv1x1+ReLU -> BatchNorm -> MaxPooling2D (evalnet.py:14-21),
// i.e. the U-Net's encoder block: its BatchNorm and pooling are applied by the consumer on load.
// Plans are imk_unet_plan objects: parameter layout, packing, optimizer state and AdamW step are the imk_unet_* calls.
#include "imk_net.h"

namespace {

struct ETopo {
    int in_c[2], in_bn[2], t_c3[2], t_c1[2], t_bn[2];   // towers A, B
    int m_c3[5], m_c1[5], m_bn[5];                      // trunk blocks
    int dense[2];
};

ETopo make_etopo(const imk_unet_plan *p) {
    ETopo t{};
    const char tw[2] = {'a', 'b'};
    char nm[16];
    for (int s = 0; s < 2; ++s) {
        snprintf(nm, sizeof nm, "%c.in.c", tw[s]); t.in_c[s] = p->find(nm);
        snprintf(nm, sizeof nm, "%c.in.bn", tw[s]); t.in_bn[s] = p->find(nm);
        snprintf(nm, sizeof nm, "%c.c3", tw[s]); t.t_c3[s] = p->find(nm);
        snprintf(nm, sizeof nm, "%c.c1", tw[s]); t.t_c1[s] = p->find(nm);
        snprintf(nm, sizeof nm, "%c.bn", tw[s]); t.t_bn[s] = p->find(nm);
    }
    for (int i = 0; i < 5; ++i) {
        snprintf(nm, sizeof nm, "m%d.c3", i + 1); t.m_c3[i] = p->find(nm);
        snprintf(nm, sizeof nm, "m%d.c1", i + 1); t.m_c1[i] = p->find(nm);
        snprintf(nm, sizeof nm, "m%d.bn", i + 1); t.m_bn[i] = p->find(nm);
    }
    t.dense[0] = p->find(p->ecfg.two_heads ? "iou" : "dense");
    t.dense[1] = p->ecfg.two_heads ? p->find("detection") : -1;
    return t;
}

// Layer (= parameter) order is Keras' creation order of evalnet.py:24-45: tower A (input block, conv block), tower B,
// the five trunk blocks, the Dense head(s).
void build_layers(imk_unet_plan *p) {
    const imk_evalnet_cfg &e = p->ecfg;
    const int F = e.ch[0];
    const char tw[2] = {'a', 'b'};
    const int cin[2] = {e.ca, e.cb};
    const int norm[2] = {e.normalize_a, e.normalize_b};
    char nm[16];
    for (int s = 0; s < 2; ++s) {
        snprintf(nm, sizeof nm, "%c.in.c", tw[s]);
        const int ic = add_conv(p, nm, 1, cin[s], F, 0);
        if (!norm[s] && !(s == 1 && e.b_onehot)) p->layers[ic].flags |= IMK_LF_U8_RAW;
        snprintf(nm, sizeof nm, "%c.in.bn", tw[s]);
        const int ib = add_bn(p, nm, F, 0, ic);
        snprintf(nm, sizeof nm, "%c.c3", tw[s]);
        const int c3 = add_conv(p, nm, 3, F, F, 0);
        snprintf(nm, sizeof nm, "%c.c1", tw[s]);
        const int c1 = add_conv(p, nm, 1, F, F, 0);
        snprintf(nm, sizeof nm, "%c.bn", tw[s]);
        add_bn(p, nm, F, 0, c1);
        if (s == 1 && e.b_onehot) set_src(p, ic, LM_RAW, IMK_SRC_ONEHOT);   // 1x1 conv over the one-hot stack
        else set_src(p, ic, LM_U8, s ? IMK_SRC_XB : IMK_SRC_XA);
        set_src(p, c3, LM_AFFINE, ic, ib);
        set_src(p, c1, LM_RAW, c3);
    }
    int prev_c1 = -1, prev_bn = -1, prev_f = 2 * F;
    for (int i = 0; i < 5; ++i) {
        const int f = e.ch[i], res = i + 1;
        snprintf(nm, sizeof nm, "m%d.c3", i + 1);
        const int c3 = add_conv(p, nm, 3, prev_f, f, res);
        snprintf(nm, sizeof nm, "m%d.c1", i + 1);
        const int c1 = add_conv(p, nm, 1, f, f, res);
        snprintf(nm, sizeof nm, "m%d.bn", i + 1);
        const int bn = add_bn(p, nm, f, res, c1);
        if (i == 0) set_src(p, c3, LM_RAW, IMK_SRC_CAT);
        else set_src(p, c3, LM_POOL, prev_c1, prev_bn);
        set_src(p, c1, LM_RAW, c3);
        prev_c1 = c1; prev_bn = bn; prev_f = f;
    }
    for (int h = 0; h < (e.two_heads ? 2 : 1); ++h) {
        const int d = add_conv(p, e.two_heads ? (h ? "detection" : "iou") : "dense", 1, e.ch[4], e.n_out, 6);
        p->layers[d].flags = IMK_LF_DENSE;
    }
    finish_layout(p);
}

Ws make_ews(const imk_unet_plan *p, int B, int mode) {
    Ws w;
    const imk_evalnet_cfg &e = p->ecfg;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up(off + bytes); return o; };
    make_ws_layers(p, B, mode, w, take);
    const size_t px1 = (size_t)B * (e.h / 2) * (e.w / 2);
    const int cat_cs = 2 * imk_pad8(e.ch[0]);
    w.cat = take(px1 * cat_cs * 2);
    if (e.b_onehot) w.onehot = take((size_t)B * e.h * e.w * imk_pad8(e.cb) * 2);
    w.probs = take((size_t)B * 2 * e.n_out * sizeof(float));   // training: the head's outputs of the batch
    if (mode == 1) {
        w.dcat = take(px1 * cat_cs * 2);
        for (int i = 0; i < 5; ++i) {   // gradient w.r.t. the pooled output of trunk block i+1, at res i+2
            const Dim d = res_dim(p->cfg, i + 2);
            w.dP[i] = take((size_t)B * d.h * d.w * imk_pad8(e.ch[i]) * 2);
        }
        w.head_partial = take(imk_evalnet_head_partial_floats(B, e.two_heads ? 2 : 1, e.n_out, e.ch[4]) * sizeof(float));
    }
    w.total = off;
    return w;
}

int run_forward(Ctx &c, const ETopo &t, float *params_rw) {
    int rc;
#define OK(e) do { rc = (e); if (rc) return rc; } while (0)
    const imk_evalnet_cfg &e = c.p->ecfg;
    if (e.b_onehot)
        OK(imk_launch_onehot(c.x_in[1], (long long)c.B * e.h * e.w, imk_pad8(e.cb), reinterpret_cast<f16 *>(c.base + c.ws.onehot),
                             c.stream));
    for (int s = 0; s < 2; ++s) {
        OK(run_conv_fwd(c, t.in_c[s], params_rw));
        OK(run_conv_pair(c, t.t_c3[s], t.t_c1[s], params_rw));
    }
    const int csF = imk_pad8(e.ch[0]);
    OK(imk_launch_concat_pool(c.act(t.t_c1[0]), c.bn_scale(t.t_bn[0]), c.bn_shift(t.t_bn[0]), csF, c.act(t.t_c1[1]),
                              c.bn_scale(t.t_bn[1]), c.bn_shift(t.t_bn[1]), csF, c.B, e.h / 2, e.w / 2,
                              reinterpret_cast<f16 *>(c.base + c.ws.cat), c.stream));
    for (int i = 0; i < 5; ++i) OK(run_conv_pair(c, t.m_c3[i], t.m_c1[i], params_rw));
#undef OK
    return IMK_OK;
}

// the tail: BN + pool of the last block, global average pool, Dense head(s); with targets also loss and gradients
int run_head(Ctx &c, const ETopo &t, float *out, const float *y, const ImkCtl *ctl, float *stats) {
    const imk_evalnet_cfg &e = c.p->ecfg;
    const int nh = e.two_heads ? 2 : 1;
    const float *w[2] = {nullptr, nullptr}, *b[2] = {nullptr, nullptr};
    for (int h = 0; h < nh; ++h) { const ImkLayer &d = c.p->layers[t.dense[h]]; w[h] = c.params + d.off_w; b[h] = c.params + d.off_b; }
    const Dim d5 = res_dim(c.p->cfg, 5);
    return imk_launch_evalnet_head(c.act(t.m_c1[4]), c.bn_scale(t.m_bn[4]), c.bn_shift(t.m_bn[4]), w, b, nh, e.n_out, e.ch[4],
                                   imk_pad8(e.ch[4]), c.B, d5.h, d5.w, out, y, ctl, stats,
                                   y ? reinterpret_cast<f16 *>(c.base + c.ws.dP[4]) : nullptr,
                                   y ? reinterpret_cast<float *>(c.base + c.ws.head_partial) : nullptr, c.stream);
}

bool ecfg_ok(const imk_evalnet_cfg *c) {
    if (!c) return false;
    // six 2x2 poolings; sizes that are not multiples of 64 lose odd last rows / columns like Keras' 'valid' pooling
    // (208 x 416: 13 -> 6 -> 3).  Full resolution itself must be even (the towers' pooled outputs are concatenated).
    if (c->h < 64 || c->w < 64 || (c->h % 2) || (c->w % 2)) return false;
    if (c->ca < 1 || c->ca > 4 || c->cb < 1 || c->cb > (c->b_onehot ? 64 : 4)) return false;   // uint8 stems of the pipelined conv kernel
    if (c->n_out < 1 || c->n_out > 64) return false;
    for (int i = 0; i < 5; ++i) if (c->ch[i] < 1 || c->ch[i] > 512) return false;
    if (c->ch[0] % 8) return false;   // the towers' channels sit side by side in the concatenated tensor
    return true;
}

}  // namespace

// =====================================================================================================
extern "C" int imk_evalnet_plan_create(const imk_evalnet_cfg *cfg, imk_unet_plan **out) {
    IMK_CHECK_ARG(out);
    if (!ecfg_ok(cfg)) return IMK_EINVAL;
    if ((long long)cfg->h * cfg->w >= imk_conv_max_pixels() || cfg->w >= (1 << 16)) return IMK_EUNSUPPORTED;   // include/imk.h: image size limit
    imk_unet_plan *p = new (std::nothrow) imk_unet_plan();
    if (!p) return IMK_EINVAL;
    p->net = 1;
    p->ecfg = *cfg;
    p->cfg = imk_unet_cfg{cfg->h, cfg->w, cfg->ca, cfg->n_out, {cfg->ch[0], cfg->ch[1], cfg->ch[2], cfg->ch[3], cfg->ch[4]}, 0};
    build_layers(p);
    *out = p;
    return IMK_OK;
}

extern "C" int64_t imk_evalnet_workspace_bytes(const imk_unet_plan *plan, int batch, int mode) {
    if (!plan || plan->net != 1 || batch <= 0 || (mode != 0 && mode != 1)) return IMK_EINVAL;
    return (int64_t)make_ews(plan, batch, mode).total;
}

extern "C" int imk_evalnet_forward(const imk_unet_plan *plan, const float *params, const void *packed, const uint8_t *xa,
                                   const uint8_t *xb, int batch, float *out, void *workspace, int64_t workspace_bytes,
                                   void *stream_) {
    IMK_CHECK_ARG(plan && plan->net == 1 && params && packed && xa && xb && out && workspace && batch > 0);
    Ctx c{plan, make_ews(plan, batch, 0), (uint8_t *)workspace, params, (const uint8_t *)packed, batch, false,
          (hipStream_t)stream_};
    if ((int64_t)c.ws.total > workspace_bytes) return IMK_EWORKSPACE;
    c.x_in[0] = xa; c.x_in[1] = xb;
    const ETopo t = make_etopo(plan);
    int rc = run_forward(c, t, nullptr);
    if (rc) return rc;
    return run_head(c, t, out, nullptr, nullptr, nullptr);
}

extern "C" int imk_evalnet_tensor_info(const imk_unet_plan *plan, int batch, int mode, int layer_idx, int which,
                                       int64_t *byte_offset, int *h, int *w, int *c, int *c_stride) {
    IMK_CHECK_ARG(plan && plan->net == 1 && batch > 0 && layer_idx >= 0 && layer_idx < (int)plan->layers.size());
    IMK_CHECK_ARG(mode == 0 || mode == 1);
    const Ws ws = make_ews(plan, batch, mode);
    const ImkLayer &l = plan->layers[layer_idx];
    IMK_CHECK_ARG(!(l.flags & IMK_LF_DENSE));
    const Dim d = res_dim(plan->cfg, l.res);
    size_t off = 0;
    if (which == 0 && l.kind == 0) off = ws.L[layer_idx].out;
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

// One training pass: forward with batch statistics, losses (head 0 mean squared error; head 1 binary cross-entropy,
// functions.py:4708 `loss=['mse', 'binary_crossentropy']`; the single head of get_evalnet: mse, functions.py:4492),
// backward.  grads [n_trainable]; stats [8]: loss, overflow flag, loss scale, step, loss of head 0, loss of head 1.
// out [B, n_heads*n_out]: the sigmoid outputs of this (training-mode) pass.
extern "C" int imk_evalnet_fwd_bwd(const imk_unet_plan *plan, float *params, void *packed, void *state, const uint8_t *xa,
                                   const uint8_t *xb, const float *y, int batch, float *out, float *grads, float *stats,
                                   void *workspace, int64_t workspace_bytes, void *stream_) {
    IMK_CHECK_ARG(plan && plan->net == 1 && params && packed && state && xa && xb && y && grads && stats && workspace);
    IMK_CHECK_ARG(batch > 0);
    hipStream_t stream = (hipStream_t)stream_;
    Ctx c{plan, make_ews(plan, batch, 1), (uint8_t *)workspace, params, (const uint8_t *)packed, batch, true, stream};
    if ((int64_t)c.ws.total > workspace_bytes) return IMK_EWORKSPACE;
    c.x_in[0] = xa; c.x_in[1] = xb;
    const StateView sv = state_view(plan, state);
    const ETopo t = make_etopo(plan);
    const imk_evalnet_cfg &e = plan->ecfg;
    const int nh = e.two_heads ? 2 : 1;
    int rc;
#define OK(e_) do { rc = (e_); if (rc) return rc; } while (0)
    OK(run_forward(c, t, params));
    float *head_out = out ? out : reinterpret_cast<float *>(c.base + c.ws.probs);
    OK(run_head(c, t, head_out, y, sv.ctl, stats));

    static const int n_side_env = []() { const char *s = getenv("IMK_SIDE_STREAMS"); int v = s ? atoi(s) : 1;
                                         return v < 0 ? 0 : (v > imk_unet_plan::MAX_SIDE ? imk_unet_plan::MAX_SIDE : v); }();
    const bool side_on = !plan->dbg_single_stream && n_side_env > 0 && ensure_side_streams(plan, n_side_env);
    Bwd b{c, grads, sv.ctl, stats + 1, side_on ? n_side_env : 0, 1LL << 62};
    ImkStopRingScope stop_ring(plan, stream, b.n_side);      // the backward pass's launches carry their own events (imk_common.h)
    {   // Dense gradients and the loss values: batch reduction of the head kernel's per-sample terms
        const ImkLayer &d0 = plan->layers[t.dense[0]];
        const ImkLayer *d1 = nh > 1 ? &plan->layers[t.dense[1]] : nullptr;
        OK(imk_launch_evalnet_head_reduce(reinterpret_cast<const float *>(c.base + c.ws.head_partial), batch, nh, e.n_out,
                                          e.ch[4], &sv.ctl->inv_loss_scale, grads + d0.off_w, grads + d0.off_b,
                                          d1 ? grads + d1->off_w : nullptr, d1 ? grads + d1->off_b : nullptr, stats + 1, stats,
                                          stream));
    }
    // trunk blocks 5..1: dy of the block's BatchNorm = max-pool scatter of the gradient of its pooled output
    for (int i = 4; i >= 0; --i) {
        OK(b.bn_bwd(t.m_bn[i], 1, nullptr, reinterpret_cast<f16 *>(c.base + c.ws.dP[i])));
        OK(b.wgrad_dgrad(t.m_c1[i], c.dA(t.m_c3[i]), c.act(t.m_c3[i])));
        f16 *dst = i > 0 ? reinterpret_cast<f16 *>(c.base + c.ws.dP[i - 1]) : reinterpret_cast<f16 *>(c.base + c.ws.dcat);
        OK(b.wgrad_dgrad(t.m_c3[i], dst, nullptr));
        OK(b.flush_wgrads());
    }
    // towers: each takes its channel slice of the concatenated gradient
    const int csF = imk_pad8(e.ch[0]);
    for (int s = 1; s >= 0; --s) {
        const f16 *dcat = reinterpret_cast<const f16 *>(c.base + c.ws.dcat) + s * csF;
        {   // IMK_EVALNET_TAIL: 0 = round 4's order; 1 = the last tower's weight gradients released with their dgrads and their split
            // reductions left to the end of the step (Bwd::tail_early, Bwd::defer_finalize); 2 (default) = both towers released early
            static const int tail = []() { const char *e = getenv("IMK_EVALNET_TAIL"); return e ? atoi(e) : 2; }();
            if (tail >= 2 || (tail == 1 && s == 0)) b.tail_early = true;
            if (tail >= 1 && s == 0) b.defer_finalize = true;
        }
        OK(b.bn_bwd(t.t_bn[s], 1, nullptr, dcat, 2 * csF));
        OK(b.wgrad_dgrad(t.t_c1[s], c.dA(t.t_c3[s]), c.act(t.t_c3[s])));
        OK(b.wgrad_dgrad(t.t_c3[s], c.dy(t.in_bn[s]), nullptr, t.in_bn[s]));
        OK(b.bn_bwd(t.in_bn[s], 0, nullptr, nullptr));
        OK(b.wgrad(t.in_c[s], nullptr, s == 0));      // the very last one stays on the main stream (Bwd::wgrad)
    }
    OK(b.finish_wgrads());
#undef OK
    return IMK_OK;
}
