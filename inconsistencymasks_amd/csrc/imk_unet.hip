// Host orchestration of the tiny U-Net (unet.py:4-67) on top of the kernels in imk_conv.hip /
// imk_elem.hip: parameter layout, weight packing, workspace layout, batched inference, ensemble
// inference + IM, and the training step (forward with batch statistics, loss, backward, AdamW).
// Everything is enqueued on the caller's stream; nothing here allocates or synchronises.
#include <cstdlib>
#include <cstring>
#include <new>
#include "imk_elem.h"
#include "imk_kernels.h"
#include "imk_plan.h"

namespace {

constexpr size_t ALIGN = 256;
inline size_t up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

// ---- topology ------------------------------------------------------------------------------------
void add_conv(imk_unet_plan *p, const char *name, int k, int cin, int cout, int res) {
    ImkLayer l{};
    l.name = name; l.kind = 0; l.ksize = k; l.cin = cin; l.cout = cout; l.res = res;
    p->layers.push_back(l);
}
void add_bn(imk_unet_plan *p, const char *name, int c, int res) {
    ImkLayer l{};
    l.name = name; l.kind = 1; l.ksize = 0; l.cin = c; l.cout = c; l.res = res;
    p->layers.push_back(l);
}

void build_layers(imk_unet_plan *p) {
    const int *ch = p->cfg.ch;  // 16a 32a 64a 128a 256a
    add_conv(p, "in.c", 1, p->cfg.c_in, ch[0], 0);
    add_bn(p, "in.bn", ch[0], 0);
    const int enc_in[4] = {ch[0], ch[0], ch[1], ch[2]};
    const int enc_f[4] = {ch[0], ch[1], ch[2], ch[3]};
    char nm[16];
    for (int i = 0; i < 4; ++i) {
        snprintf(nm, sizeof nm, "e%d.c3", i + 1); add_conv(p, nm, 3, enc_in[i], enc_f[i], i);
        snprintf(nm, sizeof nm, "e%d.c1", i + 1); add_conv(p, nm, 1, enc_f[i], enc_f[i], i);
        snprintf(nm, sizeof nm, "e%d.bn", i + 1); add_bn(p, nm, enc_f[i], i);
    }
    add_conv(p, "b.c3", 3, ch[3], ch[4], 4);
    add_conv(p, "b.c1", 1, ch[4], ch[3], 4);
    add_bn(p, "b.bn", ch[3], 4);
    const int dec_in[4] = {ch[3], ch[2], ch[1], ch[0]};
    const int dec_f1[4] = {ch[3], ch[2], ch[1], ch[0]};
    const int dec_f2[4] = {ch[2], ch[1], ch[0], ch[0]};
    for (int j = 0; j < 4; ++j) {
        const int res = 3 - j;
        snprintf(nm, sizeof nm, "d%d.ca", j + 6); add_conv(p, nm, 1, dec_in[j], dec_f1[j], res);
        snprintf(nm, sizeof nm, "d%d.bna", j + 6); add_bn(p, nm, dec_f1[j], res);
        snprintf(nm, sizeof nm, "d%d.c3", j + 6); add_conv(p, nm, 3, dec_f1[j], dec_f1[j], res);
        snprintf(nm, sizeof nm, "d%d.c1", j + 6); add_conv(p, nm, 1, dec_f1[j], dec_f2[j], res);
        snprintf(nm, sizeof nm, "d%d.bnb", j + 6); add_bn(p, nm, dec_f2[j], res);
    }
    add_conv(p, "out", 1, ch[0], p->cfg.n_out, 0);

    // flat parameter layout: trainable section, then moving statistics
    int64_t off = 0;
    for (auto &l : p->layers) {
        if (l.kind == 0) { l.off_w = off; off += (int64_t)l.ksize * l.ksize * l.cin * l.cout; l.off_b = off; off += l.cout; }
        else { l.off_w = off; off += l.cout; l.off_b = off; off += l.cout; }
        l.off_mean = l.off_var = -1;
    }
    p->n_trainable = off;
    for (auto &l : p->layers)
        if (l.kind == 1) { l.off_mean = off; off += l.cout; l.off_var = off; off += l.cout; }
    p->n_total = off;

    // packed buffer
    size_t pk = 0;
    for (auto &l : p->layers) {
        if (l.kind == 0) {
            const bool u8 = (&l == &p->layers[0]);   // the stem reads the uint8 image
            l.pk_bytes_fwd = (int64_t)imk_packed_conv_halfs(l.ksize, l.cin, l.cout, 0, imk_conv_pair_layout(l.cin, l.cout, u8)) * 2;
            l.pk_bytes_bwd = (int64_t)imk_packed_conv_halfs(l.ksize, l.cin, l.cout, 1, imk_conv_pair_layout(l.cout, l.cin, false)) * 2;
            l.pk_fwd = (int64_t)pk; pk = up(pk + l.pk_bytes_fwd);
            l.pk_bwd = (int64_t)pk; pk = up(pk + l.pk_bytes_bwd);
            l.pk_chain = -1;
            if (l.ksize == 1 && l.cin <= 16 && l.cout <= 16) { l.pk_chain = (int64_t)pk; pk = up(pk + 1024); }
        } else {
            l.pk_scale = (int64_t)pk; pk = up(pk + 2 * (size_t)imk_pad8(l.cout) * sizeof(float));
        }
    }
    p->packed_bytes = (int64_t)pk;
}

// ---- workspace -------------------------------------------------------------------------------------
struct LayerWs {
    size_t out = 0;            // conv: output tensor fp16 [B,H,W,cs]
    size_t dA = 0;             // conv (train): gradient w.r.t. pre-activation output
    size_t dy = 0;             // bn (train): gradient w.r.t. the BN output
    size_t stats_partial = 0;  // bn (train): [n_tiles][2cs]
    size_t scale = 0;          // bn (train): scale[cs], shift[cs]
    size_t save = 0;           // bn (train): mean[cs], invstd[cs]
    size_t bwd_partial = 0;    // bn (train)
    size_t coef = 0;           // bn (train): [3][cs]
    size_t wg_partial = 0;     // conv (train): this layer's weight-gradient partials (+ stage-1 scratch)
    int n_stats_tiles = 0;
    int n_bwd_rows = 0;        // bn (train): capacity of bwd_partial in rows
};

struct Ws {
    std::vector<LayerWs> L;
    size_t dU[4] = {0, 0, 0, 0};   // train: gradient w.r.t. decoder j's upsample+add output
    size_t dP[4] = {0, 0, 0, 0};   // train: gradient w.r.t. the pooled input of encoder i+1 / bottleneck
    size_t probs = 0, dlogit = 0, loss_partial = 0;
    size_t total = 0;
};

struct Dim { int h, w; };
inline Dim res_dim(const imk_unet_cfg &c, int res) { return Dim{c.h >> res, c.w >> res}; }

// index of the conv whose output a given conv / bn consumes, and friends
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

// the conv that feeds each BN
int bn_producer(const Topo &t, int bn) {
    if (bn == t.in_bn) return t.in_c;
    for (int i = 0; i < 4; ++i) if (bn == t.e_bn[i]) return t.e_c1[i];
    if (bn == t.b_bn) return t.b_c1;
    for (int j = 0; j < 4; ++j) { if (bn == t.d_bna[j]) return t.d_ca[j]; if (bn == t.d_bnb[j]) return t.d_c1[j]; }
    return -1;
}

Ws make_ws(const imk_unet_plan *p, int B, int mode) {
    Ws w;
    const Topo t = make_topo(p);
    w.L.resize(p->layers.size());
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up(off + bytes); return o; };
    const int n = (int)p->layers.size();
    for (int i = 0; i < n; ++i) {
        const ImkLayer &l = p->layers[i];
        const Dim d = res_dim(p->cfg, l.res);
        const size_t px = (size_t)B * d.h * d.w;
        if (l.kind == 0) {
            if (i != t.out) w.L[i].out = take(px * imk_pad8(l.cout) * 2);
            if (mode == 1) {
                w.L[i].dA = take(px * imk_pad8(l.cout) * 2);
                w.L[i].wg_partial = take(imk_wgrad_partial_floats(B, d.h, d.w, l.ksize, l.cin, l.cout) * sizeof(float));
            }
        } else if (mode == 1) {
            const int cs = imk_pad8(l.cout);
            const ImkLayer &pc = p->layers[bn_producer(t, i)];
            w.L[i].n_stats_tiles = imk_conv_num_tiles(B, d.h, d.w, imk_pad8(pc.cin), pc.ksize);
            w.L[i].stats_partial = take((size_t)w.L[i].n_stats_tiles * 2 * cs * sizeof(float));
            w.L[i].scale = take(2 * (size_t)cs * sizeof(float));
            w.L[i].save = take(2 * (size_t)cs * sizeof(float));
            w.L[i].n_bwd_rows = imk_bn_prep_blocks(B, d.h, d.w, cs);
            const int conv_rows = B * imk_cdiv(d.h, 8) * imk_cdiv(d.w, 16);   // most rows a dgrad epilogue can write
            if (conv_rows > w.L[i].n_bwd_rows) w.L[i].n_bwd_rows = conv_rows;
            w.L[i].bwd_partial = take((size_t)w.L[i].n_bwd_rows * 2 * cs * sizeof(float));
            w.L[i].coef = take(3 * (size_t)cs * sizeof(float));
            w.L[i].dy = take(px * cs * 2);
        }
    }
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

// ---- one forward pass --------------------------------------------------------------------------------
struct Ctx {
    const imk_unet_plan *p;
    Topo t;
    Ws ws;
    uint8_t *base;        // workspace
    const float *params;
    const uint8_t *packed;
    int B;
    bool train;
    hipStream_t stream;
    f16 *act(int conv) const { return reinterpret_cast<f16 *>(base + ws.L[conv].out); }
    f16 *dA(int conv) const { return reinterpret_cast<f16 *>(base + ws.L[conv].dA); }
    f16 *dy(int bn) const { return reinterpret_cast<f16 *>(base + ws.L[bn].dy); }
    const float *bn_scale(int bn) const {
        return train ? reinterpret_cast<const float *>(base + ws.L[bn].scale)
                     : reinterpret_cast<const float *>(packed + p->layers[bn].pk_scale);
    }
    const float *bn_shift(int bn) const { return bn_scale(bn) + imk_pad8(p->layers[bn].cout); }
    const f16 *wfwd(int conv) const { return reinterpret_cast<const f16 *>(packed + p->layers[conv].pk_fwd); }
    const f16 *wbwd(int conv) const { return reinterpret_cast<const f16 *>(packed + p->layers[conv].pk_bwd); }
};

// the input description of every conv (shared by forward and wgrad)
ImkInput conv_input(const Ctx &c, int conv, const uint8_t *x_u8) {
    const Topo &t = c.t;
    const ImkLayer &l = c.p->layers[conv];
    ImkInput in{};
    in.cin = l.cin;
    in.cs_in = imk_pad8(l.cin);
    auto affine = [&](int src_conv, int bn, int mode) {
        in.in = c.act(src_conv); in.sc = c.bn_scale(bn); in.sh = c.bn_shift(bn); in.lmode = mode;
    };
    if (conv == t.in_c) { in.in = x_u8; in.lmode = LM_U8; in.cs_in = 8; return in; }
    if (conv == t.e_c3[0]) { affine(t.in_c, t.in_bn, LM_AFFINE); return in; }
    for (int i = 1; i < 4; ++i) if (conv == t.e_c3[i]) { affine(t.e_c1[i - 1], t.e_bn[i - 1], LM_POOL); return in; }
    if (conv == t.b_c3) { affine(t.e_c1[3], t.e_bn[3], LM_POOL); return in; }
    for (int j = 0; j < 4; ++j) {
        if (conv == t.d_ca[j]) {
            const int lo_c = j == 0 ? t.b_c1 : t.d_c1[j - 1], lo_bn = j == 0 ? t.b_bn : t.d_bnb[j - 1];
            const int sk = 3 - j;
            affine(lo_c, lo_bn, LM_UPADD);
            in.in2 = c.act(t.e_c1[sk]); in.sc2 = c.bn_scale(t.e_bn[sk]); in.sh2 = c.bn_shift(t.e_bn[sk]);
            return in;
        }
        if (conv == t.d_c3[j]) { affine(t.d_ca[j], t.d_bna[j], LM_AFFINE); return in; }
        if (conv == t.d_c1[j]) { in.in = c.act(t.d_c3[j]); in.lmode = LM_RAW; return in; }
    }
    for (int i = 0; i < 4; ++i) if (conv == t.e_c1[i]) { in.in = c.act(t.e_c3[i]); in.lmode = LM_RAW; return in; }
    if (conv == t.b_c1) { in.in = c.act(t.b_c3); in.lmode = LM_RAW; return in; }
    if (conv == t.out) { affine(t.d_c1[3], t.d_bnb[3], LM_AFFINE); return in; }
    return in;
}

int bn_of_conv(const Topo &t, int conv) {  // the BN that directly follows a conv, or -1
    if (conv == t.in_c) return t.in_bn;
    for (int i = 0; i < 4; ++i) if (conv == t.e_c1[i]) return t.e_bn[i];
    if (conv == t.b_c1) return t.b_bn;
    for (int j = 0; j < 4; ++j) { if (conv == t.d_ca[j]) return t.d_bna[j]; if (conv == t.d_c1[j]) return t.d_bnb[j]; }
    return -1;
}

bool g_materialize = false;   // imk_debug_materialize(1): inference also stores the intermediates of fused kernels

// conv2 >= 0: fuse the 1x1 conv `conv2` (whose only input is conv's output) into the same kernel when possible.
// Returns 1 in *fused if it did.
int run_conv_fwd(Ctx &c, int conv, const uint8_t *x_u8, float *params_rw, int conv2 = -1, bool *fused = nullptr) {
    const ImkLayer &l = c.p->layers[conv];
    const Dim d = res_dim(c.p->cfg, l.res);
    ImkConvArgs a{};
    a.x = conv_input(c, conv, x_u8);
    a.B = c.B; a.H = d.h; a.W = d.w; a.ksize = l.ksize;
    a.cout = l.cout; a.cs_out = imk_pad8(l.cout);
    a.wpk = c.wfwd(conv);
    a.bias = c.params + l.off_b;
    a.out = c.act(conv);
    a.epi = EP_RELU;
    int stat_conv = conv;
    if (fused) *fused = false;
    if (conv2 >= 0) {
        const ImkLayer &l2 = c.p->layers[conv2];
        if (l2.pk_chain >= 0 && imk_conv_can_chain(a, l2.cout)) {
            a.wpk2 = reinterpret_cast<const f16 *>(c.packed + l2.pk_chain);
            a.bias2 = c.params + l2.off_b;
            a.out2 = c.act(conv2);
            a.cout2 = l2.cout; a.cs_out2 = imk_pad8(l2.cout);
            if (!c.train && !g_materialize) a.out = nullptr;   // the intermediate never leaves the chip
            stat_conv = conv2;
            if (fused) *fused = true;
        }
    }
    const int bn = bn_of_conv(c.t, stat_conv);
    int rows = 0;
    if (c.train && bn >= 0) {
        a.stats_partial = reinterpret_cast<float *>(c.base + c.ws.L[bn].stats_partial);
        a.stats_rows = &rows;
    }
    int rc = imk_launch_conv(a, c.stream);
    if (rc) return rc;
    if (c.train && bn >= 0) {
        const ImkLayer &b = c.p->layers[bn];
        const int cs = imk_pad8(b.cout);
        float *sc = reinterpret_cast<float *>(c.base + c.ws.L[bn].scale);
        float *sv = reinterpret_cast<float *>(c.base + c.ws.L[bn].save);
        if (rows <= 0 || rows > c.ws.L[bn].n_stats_tiles) return IMK_EWORKSPACE;
        rc = imk_launch_bn_finalize(a.stats_partial, rows, b.cout, cs, (double)c.B * d.h * d.w,
                                    c.params + b.off_w, c.params + b.off_b, params_rw + b.off_mean, params_rw + b.off_var,
                                    sc, sc + cs, sv, sv + cs, c.stream);
    }
    return rc;
}

int run_forward(Ctx &c, const uint8_t *x, float *probs, float *params_rw) {
    const Topo &t = c.t;
    int rc;
#define RUN(conv) do { rc = run_conv_fwd(c, (conv), x, params_rw); if (rc) return rc; } while (0)
    // Conv3x3+ReLU -> Conv1x1+ReLU pairs run as one kernel where the channel counts allow it
#define RUN_PAIR(c3, c1) do { bool f_ = false; rc = run_conv_fwd(c, (c3), x, params_rw, (c1), &f_); if (rc) return rc; \
                              if (!f_) RUN(c1); } while (0)
    RUN(t.in_c);
    for (int i = 0; i < 4; ++i) RUN_PAIR(t.e_c3[i], t.e_c1[i]);
    RUN_PAIR(t.b_c3, t.b_c1);
    for (int j = 0; j < 4; ++j) { RUN(t.d_ca[j]); RUN_PAIR(t.d_c3[j], t.d_c1[j]); }
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
extern "C" int imk_unet_plan_create(const imk_unet_cfg *cfg, imk_unet_plan **out) {
    IMK_CHECK_ARG(out);
    if (!cfg_ok(cfg)) return IMK_EINVAL;
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
        (void)hipStreamSynchronize(plan->side[i]);
        (void)hipStreamDestroy(plan->side[i]);
        if (plan->ev_join[i]) (void)hipEventDestroy(plan->ev_join[i]);
    }
    for (auto &e : plan->ev_fork) if (e) (void)hipEventDestroy(e);
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

extern "C" int imk_debug_materialize(int on) { g_materialize = on != 0; return IMK_OK; }

static bool g_single_stream = false;   // imk_debug_single_stream(1): no side streams (kernels run alone: exclusive timings)
extern "C" int imk_debug_single_stream(int on) { g_single_stream = on != 0; return IMK_OK; }

extern "C" int64_t imk_unet_packed_bytes(const imk_unet_plan *plan) { return plan ? plan->packed_bytes : IMK_EINVAL; }

// ctl / stats: non-null after an optimizer step -- the first packing launch then also closes the step (loss-scale and
// step-counter update), which saves a launch of its own.
static int pack_weights(const imk_unet_plan *plan, const float *params, void *packed, hipStream_t stream, ImkCtl *ctl,
                        const float *stats, bool fold_bn) {
    IMK_CHECK_ARG(plan && params && packed);
    uint8_t *pk = (uint8_t *)packed;
    const int out_idx = plan->find("out");
    ImkPackJobs pj{};
    ImkFoldJobs fj{};
    pj.ctl = ctl; pj.stats = stats;
    auto flush_pack = [&]() -> int { int rc = imk_launch_pack_jobs(pj, stream); pj.n = 0; pj.ctl = nullptr; return rc; };
    for (size_t i = 0; i < plan->layers.size(); ++i) {
        const ImkLayer &l = plan->layers[i];
        if (l.kind == 0) {
            for (int tr = 0; tr < 3; ++tr) {
                if (tr == 0 && (int)i == out_idx) continue;   // the head runs in fp32 from `params` directly
                if (tr == 2 && (l.pk_chain < 0 || (int)i == out_idx)) continue;
                f16 *dst = (f16 *)(pk + (tr == 2 ? l.pk_chain : (tr ? l.pk_bwd : l.pk_fwd)));
                const int pair = tr == 2 ? (imk_conv_pair_layout(l.cin, l.cout, false) && l.cin <= 8)
                                         : (tr ? imk_conv_pair_layout(l.cout, l.cin, false) : imk_conv_pair_layout(l.cin, l.cout, i == 0));
                pj.j[pj.n++] = ImkPackJob{params + l.off_w, dst, l.ksize, l.cin, l.cout, tr, pair};
                if (pj.n == IMK_PACK_MAX_JOBS) { int rc = flush_pack(); if (rc) return rc; }
            }
        } else {
            if (fj.n == IMK_FOLD_MAX_JOBS) return IMK_EUNSUPPORTED;
            fj.j[fj.n++] = ImkFoldJob{params + l.off_w, params + l.off_b, params + l.off_mean, params + l.off_var,
                                      (float *)(pk + l.pk_scale), l.cout, imk_pad8(l.cout)};
        }
    }
    int rc = flush_pack();
    if (rc) return rc;
    if (pj.ctl) return IMK_EINVAL;   // the step must have been closed by a packing launch
    return fold_bn ? imk_launch_bn_fold_jobs(fj, stream) : IMK_OK;
}

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
    Ctx c{plan, make_topo(plan), make_ws(plan, batch, 0), (uint8_t *)workspace, params, (const uint8_t *)packed, batch,
          false, (hipStream_t)stream_};
    if ((int64_t)c.ws.total > workspace_bytes) return IMK_EWORKSPACE;
    return run_forward(c, x, probs, nullptr);
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

// Ensemble inference + IM.  Workspace: [N][B,H,W,K] fp32 probabilities, then one model's activations.
// Side streams + fork/join events of a plan, created on first use (training: weight gradients; ensemble inference: one
// model per stream).
static bool ensure_side_streams(const imk_unet_plan *plan) {
    std::call_once(plan->side_once, [plan]() {
        bool ok = true;
        for (int i = 0; i < imk_unet_plan::MAX_SIDE; ++i) {
            ok = ok && hipStreamCreateWithFlags(&plan->side[i], hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&plan->ev_join[i], hipEventDisableTiming) == hipSuccess;
        }
        for (auto &e : plan->ev_fork) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        plan->side_ok = ok;
    });
    return plan->side_ok;
}

extern "C" int imk_unet_forward_im(const imk_unet_plan *plan, int n_models, const float *const *params,
                                   const void *const *packed, const uint8_t *x, int batch, float thr, int cmp_ge,
                                   const uint8_t *img, int block_in, int block_out, uint8_t *img_out, uint8_t *masks_out,
                                   uint8_t *im_out, int64_t *im_size, int64_t *pred_size, uint8_t *presence,
                                   void *workspace, int64_t workspace_bytes, void *stream_) {
    IMK_CHECK_ARG(plan && params && packed && x && workspace && batch > 0 && n_models > 0);
    const imk_unet_cfg &cf = plan->cfg;
    const size_t probs_one = up((size_t)batch * cf.h * cf.w * cf.n_out * sizeof(float));
    const Ws ws = make_ws(plan, batch, 0);
    if ((int64_t)(probs_one * n_models + ws.total) > workspace_bytes) return IMK_EWORKSPACE;
    // the probability stack must be contiguous [N,B,H,W,K]: only the last slab may be padded
    const size_t probs_exact = (size_t)batch * cf.h * cf.w * cf.n_out * sizeof(float);
    uint8_t *base = (uint8_t *)workspace;
    // The models are independent until the IM kernel: with room for one activation workspace per stream (the caller
    // passes probs + k * imk_unet_workspace_bytes, k <= 1 + MAX_SIDE) they run on k streams side by side -- the deep
    // layers of one model fill the gaps of the other's.  With room for one only, they run back to back.
    const int max_streams = 1 + imk_unet_plan::MAX_SIDE;
    int n_slabs = (int)(((size_t)workspace_bytes - probs_one * n_models) / ws.total);
    if (n_slabs > n_models) n_slabs = n_models;
    if (n_slabs > max_streams) n_slabs = max_streams;
    static const bool conc_off = []() { const char *e = getenv("IMK_ENSEMBLE_STREAMS"); return e && e[0] == '0'; }();
    if (n_slabs > 1 && (conc_off || g_single_stream || !ensure_side_streams(plan))) n_slabs = 1;
    hipStream_t main_stream = (hipStream_t)stream_;
    if (n_slabs > 1) {
        IMK_HIP(hipEventRecord(plan->ev_fork[0], main_stream));
        for (int s = 1; s < n_slabs; ++s) IMK_HIP(hipStreamWaitEvent(plan->side[s - 1], plan->ev_fork[0], 0));
    }
    for (int m = 0; m < n_models; ++m) {
        const int sl = m % n_slabs;
        Ctx c{plan, make_topo(plan), ws, base + probs_one * n_models + ws.total * sl, params[m], (const uint8_t *)packed[m],
              batch, false, sl == 0 ? main_stream : plan->side[sl - 1]};
        int rc = run_forward(c, x, (float *)(base + probs_exact * m), nullptr);
        if (rc) return rc;
    }
    for (int s = 1; s < n_slabs; ++s) {
        IMK_HIP(hipEventRecord(plan->ev_join[s - 1], plan->side[s - 1]));
        IMK_HIP(hipStreamWaitEvent(main_stream, plan->ev_join[s - 1], 0));
    }
    if (cf.act_out == 0)
        return imk_im_binary((const float *)base, n_models, batch, cf.h, cf.w, cf.n_out, thr, cmp_ge, img, cf.c_in, block_in,
                             block_out, img_out, masks_out, im_out, im_size, pred_size, stream_);
    return imk_im_multiclass((const float *)base, n_models, batch, cf.h, cf.w, cf.n_out, img, cf.c_in, block_in, block_out,
                             img_out, masks_out, im_out, im_size, presence, stream_);
}

// =====================================================================================================
// training
// =====================================================================================================
namespace {
struct StateView { float *m, *v; ImkCtl *ctl; };
StateView state_view(const imk_unet_plan *p, void *state) {
    uint8_t *b = (uint8_t *)state;
    const size_t n = up((size_t)p->n_trainable * sizeof(float));
    return StateView{(float *)b, (float *)(b + n), (ImkCtl *)(b + 2 * n)};
}

struct Bwd {
    Ctx &c;
    const uint8_t *x;
    float *grads;
    ImkCtl *ctl;
    float *found_inf;       // = stats + 1: set to 1 by any gradient kernel that sees a non-finite value
    int n_side;             // side streams in use (0: everything on c.stream); weight-gradient work is dealt round-robin
    long long side_max_pixels;   // layers with at most this many pixels run their wgrad on a side stream
    ImkWgFinalJobs jobs{};
    int n_fork = 0;
    bool used_side[imk_unet_plan::MAX_SIDE] = {};

    int dy_rows[64] = {};   // per BN: statistics rows written by the kernel that produced dy (0 = none, run the prep pass)

    // Where the pre-activation gradient of `conv` comes from: convs that feed a BatchNorm get it on load from that
    // BN's (dy, z, coefficients); the 3x3 convs get the materialised, ReLU-masked dgrad output of the following 1x1.
    void grad_input(int conv, ImkInput &in) const {
        const ImkLayer &l = c.p->layers[conv];
        in.cin = l.cout; in.cs_in = imk_pad8(l.cout);
        const int bn = bn_of_conv(c.t, conv);
        if (bn >= 0) {
            in.in = c.dy(bn); in.in2 = c.act(conv); in.lmode = LM_BNBWD;
            in.sc = reinterpret_cast<const float *>(c.base + c.ws.L[bn].coef);
        } else {
            in.in = c.dA(conv); in.lmode = LM_RAW;
        }
    }
    // dgrad of `conv` -> dst, optionally masked by the ReLU of the tensor `mask`.  If dst is the output gradient of
    // a BatchNorm whose only gradient source this is (stat_bn >= 0), the kernel also emits that BN's backward
    // statistics (sum dy, sum dy*z), which saves the separate reduction pass.
    void dgrad_args(int conv, f16 *dst, const f16 *mask, int stat_bn, int *rows, ImkConvArgs &a) const {
        const ImkLayer &l = c.p->layers[conv];
        const Dim d = res_dim(c.p->cfg, l.res);
        grad_input(conv, a.x);
        a.B = c.B; a.H = d.h; a.W = d.w; a.ksize = l.ksize;
        a.cout = l.cin; a.cs_out = imk_pad8(l.cin);
        a.wpk = c.wbwd(conv);
        a.out = dst;
        a.mask = mask;
        a.epi = mask ? EP_MASK : EP_PLAIN;
        if (stat_bn >= 0) {
            a.dystat_z = c.act(bn_producer(c.t, stat_bn));
            a.stats_partial = reinterpret_cast<float *>(c.base + c.ws.L[stat_bn].bwd_partial);
            a.stats_rows = rows;
        }
    }
    int dgrad_done(int stat_bn, int rows) {
        if (stat_bn >= 0) {
            if (rows <= 0 || rows > c.ws.L[stat_bn].n_bwd_rows) return IMK_EWORKSPACE;
            dy_rows[stat_bn] = rows;
        }
        return IMK_OK;
    }
    int dgrad(int conv, f16 *dst, const f16 *mask, int stat_bn = -1) {
        ImkConvArgs a{};
        int rows = 0;
        dgrad_args(conv, dst, mask, stat_bn, &rows, a);
        int rc = imk_launch_conv(a, c.stream);
        if (rc) return rc;
        return dgrad_done(stat_bn, rows);
    }
    void wgrad_args(int conv, const f16 *dA_override, ImkWgradArgs &a) const {
        const ImkLayer &l = c.p->layers[conv];
        const Dim d = res_dim(c.p->cfg, l.res);
        a.x = conv_input(c, conv, x);
        if (dA_override) {
            a.dA = dA_override;
        } else {
            ImkInput gi{};
            grad_input(conv, gi);
            a.dA = reinterpret_cast<const f16 *>(gi.in);
            if (gi.lmode == LM_BNBWD) { a.dA_z = reinterpret_cast<const f16 *>(gi.in2); a.dA_coef = gi.sc; }
        }
        a.B = c.B; a.H = d.h; a.W = d.w; a.ksize = l.ksize; a.cout = l.cout; a.cs_out = imk_pad8(l.cout);
        a.partial = reinterpret_cast<float *>(c.base + c.ws.L[conv].wg_partial);
        a.n_split = imk_wgrad_splits(c.B, d.h, d.w, l.cin, l.cout);
    }
    int wgrad_job(int conv, const ImkWgradArgs &a) {
        const ImkLayer &l = c.p->layers[conv];
        return imk_wgf_add_job(jobs, a.partial, a.n_split, l.ksize, l.cin, l.cout, grads + l.off_w, grads + l.off_b);
    }
    // Weight/bias gradient of `conv`: depends only on dA[conv] (just produced on the main stream) and on forward
    // tensors, and nothing downstream in the backward pass depends on it -> it goes to the side stream.  Forking costs
    // an event record on the main stream (a barrier packet: ~6 us before the next kernel starts), so the weight
    // gradients of a whole resolution block are queued and forked together (flush_wgrads: 11 forks per step, not 24).
    struct Pending { int conv; const f16 *dA_override; };
    Pending pending[8];
    int n_pending = 0;
    int launch_wgrad(int conv, const f16 *dA_override, hipStream_t ws) {
        ImkWgradArgs a{};
        wgrad_args(conv, dA_override, a);
        int rc = imk_launch_wgrad(a, ws);
        if (rc) return rc;
        return wgrad_job(conv, a);
    }
    int wgrad(int conv, const f16 *dA_override = nullptr) {
        const ImkLayer &l = c.p->layers[conv];
        const Dim d = res_dim(c.p->cfg, l.res);
        const bool small = (long long)c.B * d.h * d.w <= side_max_pixels;
        if (n_side > 0 && small) {
            if (n_pending == 8) { int rc = flush_wgrads(); if (rc) return rc; }
            pending[n_pending++] = Pending{conv, dA_override};
            // full-resolution layers: fork at once -- their kernels are long (the bubble is small against them) and the
            // last block's weight gradients would otherwise all start after the main chain has ended
            return l.res == 0 ? flush_wgrads() : IMK_OK;
        }
        return launch_wgrad(conv, dA_override, c.stream);
    }
    int flush_wgrads() {
        if (n_pending == 0) return IMK_OK;
        const int si = n_fork % n_side;
        hipStream_t ws = c.p->side[si];
        hipEvent_t ev = c.p->ev_fork[n_fork++];
        IMK_HIP(hipEventRecord(ev, c.stream));
        IMK_HIP(hipStreamWaitEvent(ws, ev, 0));
        used_side[si] = true;
        for (int i = 0; i < n_pending; ++i) {
            int rc = launch_wgrad(pending[i].conv, pending[i].dA_override, ws);
            if (rc) return rc;
        }
        n_pending = 0;
        // ... and their split reductions right behind them, so that only the last layers' are left for the end of the step
        int rc = imk_launch_wgrad_finalize_jobs(jobs, &ctl->inv_loss_scale, found_inf, ws);
        jobs = ImkWgFinalJobs{};
        return rc;
    }
    // wgrad (side stream) + dgrad of `conv`; both read its pre-activation gradient.  Running the two as ONE launch
    // (dgrad and wgrad blocks side by side in one grid) was measured for the deep layers: no gain (1.451 vs 1.452 ms per
    // step) -- the common LDS footprint of the fused kernel leaves one workgroup per CU.
    int wgrad_dgrad(int conv, f16 *dst, const f16 *mask, int stat_bn = -1) {
        int rc = wgrad(conv);
        if (rc) return rc;
        return dgrad(conv, dst, mask, stat_bn);
    }
    // all layers' partials -> gradients (2 launches), then join the side stream back into the main one
    int finish_wgrads() {
        // what is still queued (the stem's weight gradient) has nothing left to overlap with: main stream
        for (int i = 0; i < n_pending; ++i) {
            int rc = launch_wgrad(pending[i].conv, pending[i].dA_override, c.stream);
            if (rc) return rc;
        }
        n_pending = 0;
        for (int si = 0; si < n_side; ++si) {   // join: the reductions below read the side streams' partials
            if (!used_side[si]) continue;
            IMK_HIP(hipEventRecord(c.p->ev_join[si], c.p->side[si]));
            IMK_HIP(hipStreamWaitEvent(c.stream, c.p->ev_join[si], 0));
        }
        return imk_launch_wgrad_finalize_jobs(jobs, &ctl->inv_loss_scale, found_inf, c.stream);
    }
    // BN backward for `bn`: per-channel coefficients of  dz = A*dy + B*z + C  (+ gamma/beta gradients).  The
    // reduction (sum dy, sum dy*z) comes from the kernel that produced dy when there was exactly one (dy_rows),
    // else from a pass that also assembles dy from its sources (mode 1: skip gradient + max-pool scatter,
    // mode 2: 2x2 sum of the upsampled branch).  The consumers apply the coefficients on load (LM_BNBWD).
    int bn_bwd(int bn, int mode, const f16 *g_direct, const f16 *g_other) {
        const ImkLayer &b = c.p->layers[bn];
        const Dim d = res_dim(c.p->cfg, b.res);
        const int cs = imk_pad8(b.cout);
        const int prod = bn_producer(c.t, bn);
        const LayerWs &lw = c.ws.L[bn];
        float *partial = reinterpret_cast<float *>(c.base + lw.bwd_partial);
        float *coef = reinterpret_cast<float *>(c.base + lw.coef);
        const float *save = reinterpret_cast<const float *>(c.base + lw.save);
        const f16 *z = c.act(prod);
        int rows = dy_rows[bn];
        if (mode != 0 || rows == 0) {
            int rc = imk_launch_bn_bwd_prep(mode, mode == 0 ? c.dy(bn) : g_direct, g_other, z, c.bn_scale(bn), c.bn_shift(bn),
                                            c.dy(bn), partial, c.B, d.h, d.w, cs, c.stream);
            if (rc) return rc;
            rows = imk_bn_prep_blocks(c.B, d.h, d.w, cs);
        }
        return imk_launch_bn_bwd_coef(partial, rows, b.cout, cs, (double)c.B * d.h * d.w, c.params + b.off_w, save, save + cs,
                                      &ctl->inv_loss_scale, coef, grads + b.off_w, grads + b.off_b, found_inf, c.stream);
    }
};
}  // namespace

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
    Ctx c{plan, make_topo(plan), make_ws(plan, batch, 1), (uint8_t *)workspace, params, (const uint8_t *)packed, batch, true,
          stream};
    if ((int64_t)c.ws.total > workspace_bytes) return IMK_EWORKSPACE;
    const StateView sv = state_view(plan, state);
    const imk_unet_cfg &cf = plan->cfg;
    const Topo &t = c.t;
    const long long n_pix = (long long)batch * cf.h * cf.w;
    int rc;
#define OK(e) do { rc = (e); if (rc) return rc; } while (0)
    OK(run_forward(c, x, nullptr, params));

    // head + loss + d(loss*scale)/d(logits) in one pass over the last activation
    f16 *dlogit = reinterpret_cast<f16 *>(c.base + c.ws.dlogit);
    float *loss_partial = reinterpret_cast<float *>(c.base + c.ws.loss_partial);
    {
        const ImkLayer &o = plan->layers[t.out];
        const int bn = t.d_bnb[3];
        OK(imk_launch_head_loss(c.act(t.d_c1[3]), c.bn_scale(bn), c.bn_shift(bn), params + o.off_w, params + o.off_b, o.cin,
                                imk_pad8(o.cin), o.cout, cf.act_out, n_pix, y, sv.ctl, stats, dlogit, loss_partial, stream));
    }

    ensure_side_streams(plan);
    // The weight-gradient kernels run on a side stream, forked after the kernel that produced their gradient operand and
    // joined before the final reduction: nothing on the backward chain depends on them, and they fill the gaps that the
    // latency-bound kernels of the chain leave (1.280 vs 1.365 ms per step with all 24 on the side stream; forking only
    // the deep layers' was neutral).  IMK_SIDE_PIXELS = largest B*H*W that is forked (0: single stream).
    static const long long side_px = []() { const char *e = getenv("IMK_SIDE_PIXELS"); return e ? atoll(e) : (1LL << 62); }();
    static const int n_side_env = []() { const char *e = getenv("IMK_SIDE_STREAMS"); int v = e ? atoi(e) : 1;
                                         return v < 0 ? 0 : (v > imk_unet_plan::MAX_SIDE ? imk_unet_plan::MAX_SIDE : v); }();
    Bwd b{c, x, grads, sv.ctl, stats + 1, (plan->side_ok && side_px > 0 && !g_single_stream) ? n_side_env : 0, side_px};
    // head: its "dA" is dlogit
    OK(b.wgrad(t.out, dlogit));
    // the loss value only needs head_loss_kernel's partials: its reduction rides on the side stream, behind the head's
    // weight gradient, instead of sitting at the end of the step
    const bool loss_on_side = b.n_side > 0 && b.n_fork > 0;
    if (loss_on_side)
        OK(imk_launch_loss_finalize(loss_partial, n_pix, cf.n_out, loss_kind, stats, plan->side[(b.n_fork - 1) % b.n_side]));
    {
        const ImkLayer &l = plan->layers[t.out];
        ImkConvArgs a{};
        a.x.in = dlogit; a.x.lmode = LM_RAW; a.x.cin = l.cout; a.x.cs_in = imk_pad8(l.cout);
        a.B = batch; a.H = cf.h; a.W = cf.w; a.ksize = 1; a.cout = l.cin; a.cs_out = imk_pad8(l.cin);
        a.wpk = c.wbwd(t.out); a.out = c.dy(t.d_bnb[3]); a.epi = EP_PLAIN;
        int rows = 0;
        a.dystat_z = c.act(t.d_c1[3]);
        a.stats_partial = reinterpret_cast<float *>(c.base + c.ws.L[t.d_bnb[3]].bwd_partial);
        a.stats_rows = &rows;
        OK(imk_launch_conv(a, stream));
        if (rows <= 0 || rows > c.ws.L[t.d_bnb[3]].n_bwd_rows) return IMK_EWORKSPACE;
        b.dy_rows[t.d_bnb[3]] = rows;
    }
    // decoders 9..6
    for (int j = 3; j >= 0; --j) {
        f16 *dU = reinterpret_cast<f16 *>(c.base + c.ws.dU[j]);
        if (j == 3) OK(b.bn_bwd(t.d_bnb[j], 0, nullptr, nullptr));
        else OK(b.bn_bwd(t.d_bnb[j], 2, nullptr, reinterpret_cast<f16 *>(c.base + c.ws.dU[j + 1])));
        OK(b.wgrad_dgrad(t.d_c1[j], c.dA(t.d_c3[j]), c.act(t.d_c3[j])));
        OK(b.wgrad_dgrad(t.d_c3[j], c.dy(t.d_bna[j]), nullptr, t.d_bna[j]));
        OK(b.bn_bwd(t.d_bna[j], 0, nullptr, nullptr));
        OK(b.wgrad_dgrad(t.d_ca[j], dU, nullptr));
        OK(b.flush_wgrads());   // the block's weight gradients (and the head's, for the first block) -> side stream
    }
    // bottleneck: dy = 2x2 sum of dU[decoder 6]
    OK(b.bn_bwd(t.b_bn, 2, nullptr, reinterpret_cast<f16 *>(c.base + c.ws.dU[0])));
    OK(b.wgrad_dgrad(t.b_c1, c.dA(t.b_c3), c.act(t.b_c3)));
    OK(b.wgrad_dgrad(t.b_c3, reinterpret_cast<f16 *>(c.base + c.ws.dP[3]), nullptr));
    OK(b.flush_wgrads());
    // encoders 4..1: dy = skip gradient (dU of decoder 6+(3-i)) + max-pool scatter of dP[i]
    for (int i = 3; i >= 0; --i) {
        OK(b.bn_bwd(t.e_bn[i], 1, reinterpret_cast<f16 *>(c.base + c.ws.dU[3 - i]),
                    reinterpret_cast<f16 *>(c.base + c.ws.dP[i])));
        OK(b.wgrad_dgrad(t.e_c1[i], c.dA(t.e_c3[i]), c.act(t.e_c3[i])));
        f16 *dst = i > 0 ? reinterpret_cast<f16 *>(c.base + c.ws.dP[i - 1]) : c.dy(t.in_bn);
        OK(b.wgrad_dgrad(t.e_c3[i], dst, nullptr, i > 0 ? -1 : t.in_bn));
        OK(b.flush_wgrads());
    }
    OK(b.bn_bwd(t.in_bn, 0, nullptr, nullptr));
    OK(b.wgrad(t.in_c));
    OK(b.finish_wgrads());
    if (!loss_on_side) OK(imk_launch_loss_finalize(loss_partial, n_pix, cf.n_out, loss_kind, stats, stream));
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
