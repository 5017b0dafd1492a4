// Host orchestration shared by the two networks of the reference (imk_unet.hip: unet.py:4-67, imk_evalnet.hip:
// evalnet.py:4-47), both stacks of Conv3x3+ReLU -> Conv1x1+ReLU -> BatchNorm blocks: flat parameter layout, weight
// packing, per-layer workspace, one conv launch with its BatchNorm statistics, and the backward helpers (dgrad, weight
// gradients on a side stream, BatchNorm backward).  The network files add the topology and the order of the launches.
// Everything is enqueued on the caller's stream; nothing here allocates or synchronises.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include "imk_elem.h"
#include "imk_kernels.h"
#include "imk_plan.h"


namespace {

constexpr size_t ALIGN = 256;
inline size_t up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

// ---- layers ----------------------------------------------------------------------------------------
inline int add_conv(imk_unet_plan *p, const char *name, int k, int cin, int cout, int res) {
    ImkLayer l{};
    l.name = name; l.kind = 0; l.ksize = k; l.cin = cin; l.cout = cout; l.res = res;
    p->layers.push_back(l);
    return (int)p->layers.size() - 1;
}
inline int add_bn(imk_unet_plan *p, const char *name, int c, int res, int producer) {
    ImkLayer l{};
    l.name = name; l.kind = 1; l.ksize = 0; l.cin = c; l.cout = c; l.res = res; l.producer = producer;
    p->layers.push_back(l);
    const int idx = (int)p->layers.size() - 1;
    p->layers[producer].bn_after = idx;
    return idx;
}
inline void set_src(imk_unet_plan *p, int conv, int lmode, int src, int src_bn = -1, int src2 = -1, int src2_bn = -1) {
    ImkLayer &l = p->layers[conv];
    l.lmode = lmode; l.src = src; l.src_bn = src_bn; l.src2 = src2; l.src2_bn = src2_bn;
}

// flat parameter layout (trainable section, then moving statistics) and the packed-weight buffer
inline void finish_layout(imk_unet_plan *p) {
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

    size_t pk = 0;
    for (auto &l : p->layers) {
        l.pk_fwd = l.pk_bwd = l.pk_chain = -1;
        if (l.kind == 0) {
            if (l.flags & IMK_LF_DENSE) continue;   // fp32 only
            const bool u8 = l.lmode == LM_U8;   // a stem reads the uint8 image
            l.pk_bytes_fwd = (int64_t)imk_packed_conv_halfs(l.ksize, l.cin, l.cout, 0, imk_conv_pair_layout(l.cin, l.cout, u8)) * 2;
            l.pk_bytes_bwd = (int64_t)imk_packed_conv_halfs(l.ksize, l.cin, l.cout, 1, imk_conv_pair_layout(l.cout, l.cin, false)) * 2;
            l.pk_fwd = (int64_t)pk; pk = up(pk + l.pk_bytes_fwd);
            l.pk_bwd = (int64_t)pk; pk = up(pk + l.pk_bytes_bwd);
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
    // U-Net
    size_t dU[4] = {0, 0, 0, 0};   // train: gradient w.r.t. decoder j's upsample+add output
    size_t dP[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // train: gradient w.r.t. the pooled output of a block (U-Net: encoder i+1;
                                               // EvalNet: trunk block i+1, [5] = the last block's, written by the head)
    size_t probs = 0, dlogit = 0, loss_partial = 0;
    // EvalNet
    size_t cat = 0, dcat = 0;      // concatenated pooled tower outputs [B,H/2,W/2,2F] and their gradient
    size_t onehot = 0;             // b_onehot: input B as fp16 one-hot [B,H,W,pad8(cb)]
    size_t head_partial = 0;       // train: per-sample Dense gradients and loss terms
    size_t sched = 0;              // inference (U-Net): IMK_SCHED_BYTES of tile counters per layer (ImkConvArgs::sched), zeroed per forward
    bool has_sched = false;
    size_t total = 0;
};

struct Dim { int h, w; };
inline Dim res_dim(const imk_unet_cfg &c, int res) { return Dim{c.h >> res, c.w >> res}; }

// the per-layer part of the workspace: activations, and in training the gradients / statistics of every layer
template <typename Take>
inline void make_ws_layers(const imk_unet_plan *p, int B, int mode, Ws &w, Take &&take) {
    const int n = (int)p->layers.size();
    w.L.resize(n);
    for (int i = 0; i < n; ++i) {
        const ImkLayer &l = p->layers[i];
        const Dim d = res_dim(p->cfg, l.res);
        const size_t px = (size_t)B * d.h * d.w;
        if (l.kind == 0) {
            if (l.flags & IMK_LF_DENSE) continue;
            if (!(l.flags & IMK_LF_HEAD)) w.L[i].out = take(px * imk_pad8(l.cout) * 2);
            if (mode == 1) {
                w.L[i].dA = take(px * imk_pad8(l.cout) * 2);
                w.L[i].wg_partial = take(imk_wgrad_partial_floats(B, d.h, d.w, l.ksize, l.cin, l.cout) * sizeof(float));
            }
        } else if (mode == 1) {
            const int cs = imk_pad8(l.cout);
            const ImkLayer &pc = p->layers[l.producer];
            w.L[i].n_stats_tiles = imk_conv_num_tiles(B, d.h, d.w, imk_pad8(pc.cin), pc.ksize);
            w.L[i].stats_partial = take((size_t)w.L[i].n_stats_tiles * 2 * cs * sizeof(float));
            w.L[i].scale = take(2 * (size_t)cs * sizeof(float));
            w.L[i].save = take(2 * (size_t)cs * sizeof(float));
            w.L[i].n_bwd_rows = imk_bn_prep_blocks(B, d.h, d.w, cs);
            const int conv_rows = B * imk_cdiv(d.h, 8) * imk_cdiv(d.w, 16);   // most rows a dgrad epilogue can write
            if (conv_rows > w.L[i].n_bwd_rows) w.L[i].n_bwd_rows = conv_rows;
            // ... or the persistent 1x1 dgrad one level up, when it assembles this BatchNorm's dy (ImkConvArgs::sum2_out): one row per workgroup
            const int up_rows = std::min(imk_conv_fused_wgrad_rows_max(), B * imk_cdiv(2 * d.h, 16) * imk_cdiv(2 * d.w, 16));
            if (up_rows > w.L[i].n_bwd_rows) w.L[i].n_bwd_rows = up_rows;
            w.L[i].bwd_partial = take((size_t)w.L[i].n_bwd_rows * 2 * cs * sizeof(float));
            w.L[i].coef = take(3 * (size_t)cs * sizeof(float));
            w.L[i].dy = take(px * cs * 2);
        }
    }
}

// ---- one pass ----------------------------------------------------------------------------------------
struct Ctx {
    const imk_unet_plan *p;
    Ws ws;
    uint8_t *base;        // workspace
    const float *params;
    const uint8_t *packed;
    int B;
    bool train;
    hipStream_t stream;
    const uint8_t *x_in[2] = {nullptr, nullptr};   // the uint8 network inputs (U-Net: [0] only)
    unsigned *sched(int conv) const {     // the launch's tile counters (inference with a workspace that carries them), else null
        return (!train && ws.has_sched) ? reinterpret_cast<unsigned *>(base + ws.sched + IMK_SCHED_BYTES * (size_t)conv) : nullptr;
    }
    int ovr_conv = -1;    // ensemble inference: this conv's output lives outside the (shared) activation workspace ...
    f16 *ovr_out = nullptr;   // ... here, so that it survives until the fused head + IM kernel has read every model's
    f16 *act(int conv) const { return conv == ovr_conv ? ovr_out : reinterpret_cast<f16 *>(base + ws.L[conv].out); }
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

// the input description of every conv (shared by forward and wgrad), from the plan's graph
inline ImkInput conv_input(const Ctx &c, int conv) {
    const ImkLayer &l = c.p->layers[conv];
    ImkInput in{};
    in.cin = l.cin;
    in.cs_in = imk_pad8(l.cin);
    in.lmode = l.lmode;
    if (l.lmode == LM_U8) {
        in.in = c.x_in[l.src == IMK_SRC_XB ? 1 : 0];
        in.cs_in = 8;
        in.u8_div = (l.flags & IMK_LF_U8_RAW) ? 1.0f : 255.0f;
        return in;
    }
    in.in = l.src == IMK_SRC_CAT ? reinterpret_cast<const void *>(c.base + c.ws.cat)
          : l.src == IMK_SRC_ONEHOT ? reinterpret_cast<const void *>(c.base + c.ws.onehot) : c.act(l.src);
    if (l.src_bn >= 0) { in.sc = c.bn_scale(l.src_bn); in.sh = c.bn_shift(l.src_bn); }
    if (l.lmode == LM_POOL) { const Dim ds = res_dim(c.p->cfg, c.p->layers[l.src].res); in.src_h = ds.h; in.src_w = ds.w; }
    if (l.src2 >= 0) { in.in2 = c.act(l.src2); in.sc2 = c.bn_scale(l.src2_bn); in.sh2 = c.bn_shift(l.src2_bn); }
    return in;
}

// conv2 >= 0: fuse the 1x1 conv `conv2` (whose only input is conv's output) into the same kernel when possible.
// Returns 1 in *fused if it did.
// x_override: another description of the same input (LM_STEM: the input block computed on load); only valid together with
// a successful chain, otherwise nothing is launched and IMK_EUNSUPPORTED comes back (the caller runs the plain path).
inline int run_conv_fwd(Ctx &c, int conv, float *params_rw, int conv2 = -1, bool *fused = nullptr,
                        const ImkInput *x_override = nullptr) {
    const ImkLayer &l = c.p->layers[conv];
    const Dim d = res_dim(c.p->cfg, l.res);
    ImkConvArgs a{};
    a.x = x_override ? *x_override : conv_input(c, conv);
    a.B = c.B; a.H = d.h; a.W = d.w; a.ksize = l.ksize;
    a.cout = l.cout; a.cs_out = imk_pad8(l.cout);
    a.wpk = c.wfwd(conv);
    a.bias = c.params + l.off_b;
    a.out = c.act(conv);
    a.epi = EP_RELU;
    a.sched = c.sched(conv);
    int stat_conv = conv;
    if (fused) *fused = false;
    if (conv2 >= 0) {
        const ImkLayer &l2 = c.p->layers[conv2];
        const bool pipe_chain = l2.pk_chain >= 0 && imk_conv_can_chain(a, l2.cout);
        if (pipe_chain || (!x_override && l2.ksize == 1 && imk_conv_can_chain_tile(a, l2.cout, c.train || c.p->dbg_materialize))) {
            a.wpk2 = pipe_chain ? reinterpret_cast<const f16 *>(c.packed + l2.pk_chain) : c.wfwd(conv2);
            a.bias2 = c.params + l2.off_b;
            a.out2 = c.act(conv2);
            a.cout2 = l2.cout; a.cs_out2 = imk_pad8(l2.cout);
            if (!c.train && !c.p->dbg_materialize) a.out = nullptr;   // the intermediate never leaves the chip
            stat_conv = conv2;
            if (fused) *fused = true;
        }
    }
    if (x_override && !(a.wpk2 && !a.out)) return IMK_EUNSUPPORTED;
    const int bn = c.p->layers[stat_conv].bn_after;
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
                                    sc, sc + cs, sv, sv + cs, c.stream, c.p->bn_momentum);
    }
    return rc;
}

// Inference: Conv1x1+ReLU -> BN -> Conv3x3+ReLU -> Conv1x1+ReLU in ONE launch (conv_pipe_kernel<..., PRE>): `pre` is a decoder
// block's Conv1x1 on upsample + skip (unet.py:32-35; the input block in front of the first encoder block has its own form,
// LM_STEM).  Neither pre's output nor the 3x3's leaves the chip.  Returns IMK_EUNSUPPORTED (nothing launched) where that does not apply.
inline int run_conv_pre_pair(Ctx &c, int pre, int c3, int c1) {
    if (c.train || c.p->dbg_materialize) return IMK_EUNSUPPORTED;
    const ImkLayer &lp = c.p->layers[pre], &l = c.p->layers[c3], &l2 = c.p->layers[c1];
    if (lp.ksize != 1 || lp.bn_after < 0 || l.src != pre || l.src_bn != lp.bn_after || l.lmode != LM_AFFINE || l2.pk_chain < 0)
        return IMK_EUNSUPPORTED;
    const Dim d = res_dim(c.p->cfg, l.res);
    ImkConvArgs a{};
    a.x = conv_input(c, pre);
    a.B = c.B; a.H = d.h; a.W = d.w; a.ksize = l.ksize;
    a.cout = l.cout; a.cs_out = imk_pad8(l.cout);
    a.wpk = c.wfwd(c3);
    a.bias = c.params + l.off_b;
    a.out = nullptr;
    a.epi = EP_RELU;
    a.sched = c.sched(c3);
    a.wpk2 = reinterpret_cast<const f16 *>(c.packed + l2.pk_chain);
    a.bias2 = c.params + l2.off_b;
    a.out2 = c.act(c1);
    a.cout2 = l2.cout; a.cs_out2 = imk_pad8(l2.cout);
    {   // the chain itself must be possible with THIS conv's own input (what imk_conv_can_chain checks)
        ImkConvArgs plain = a;
        plain.x = conv_input(c, c3);
        if (!imk_conv_can_chain(plain, l2.cout)) return IMK_EUNSUPPORTED;
    }
    if (!imk_conv_can_prestage(a, lp.lmode, lp.cin, lp.cout)) return IMK_EUNSUPPORTED;
#ifdef IMK_PRE_DEBUG
    a.out = c.act(c3);                              // probe build: first-stage dumps land in the (otherwise unused) d.c3 / d.ca tensors
    a.mask = c.act(pre);
#endif
    a.pre_wpk = c.wfwd(pre);
    a.pre_bias = c.params + lp.off_b;
    a.pre_sc = c.bn_scale(lp.bn_after); a.pre_sh = c.bn_shift(lp.bn_after);
    a.pre_cout = lp.cout;
    return imk_launch_conv(a, c.stream);
}

// Conv3x3+ReLU -> Conv1x1+ReLU of a block: one kernel where the channel counts allow it, else two
inline int run_conv_pair(Ctx &c, int c3, int c1, float *params_rw) {
    bool f = false;
    int rc = run_conv_fwd(c, c3, params_rw, c1, &f);
    if (rc || f) return rc;
    return run_conv_fwd(c, c1, params_rw);
}

// ---- training ----------------------------------------------------------------------------------------
struct StateView { float *m, *v; ImkCtl *ctl; };
inline StateView state_view(const imk_unet_plan *p, void *state) {
    uint8_t *b = (uint8_t *)state;
    const size_t n = up((size_t)p->n_trainable * sizeof(float));
    return StateView{(float *)b, (float *)(b + n), (ImkCtl *)(b + 2 * n)};
}

struct Bwd {
    Ctx &c;
    float *grads;
    ImkCtl *ctl;
    float *found_inf;       // = stats + 1: set to 1 by any gradient kernel that sees a non-finite value
    int n_side;             // side streams in use (0: everything on c.stream); weight-gradient work is dealt round-robin
    long long side_max_pixels;   // layers with at most this many pixels run their wgrad on a side stream
    ImkWgFinalJobs jobs{};
    int n_fork = 0;
    bool used_side[imk_unet_plan::MAX_SIDE] = {};

    int dy_rows[64] = {};   // per BN: statistics rows written by the kernel that produced dy (0 = none, run the prep pass)
    // The BatchNorm whose output gradient is the 2x2 sum of the NEXT wgrad_dgrad call's output (a decoder's first conv: its dgrad
    // yields the gradient of upsample + add): where the pipelined kernel takes that dgrad it assembles the sums and their
    // statistics itself (ImkConvArgs::sum2_out) and bn_bwd(mode 2) skips its pass.  Consumed (reset) by that call.
    int sum2_bn = -1;

    // Where the pre-activation gradient of `conv` comes from: convs that feed a BatchNorm get it on load from that
    // BN's (dy, z, coefficients); the 3x3 convs get the materialised, ReLU-masked dgrad output of the following 1x1.
    void grad_input(int conv, ImkInput &in) const {
        const ImkLayer &l = c.p->layers[conv];
        in.cin = l.cout; in.cs_in = imk_pad8(l.cout);
        const int bn = l.bn_after;
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
            a.dystat_z = c.act(c.p->layers[stat_bn].producer);
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
    int dgrad(int conv, f16 *dst, const f16 *mask, int stat_bn = -1, int s2_bn = -1) {
        ImkConvArgs a{};
        int rows = 0;
        dgrad_args(conv, dst, mask, stat_bn, &rows, a);
        bool s2 = false;
        if (s2_bn >= 0 && stat_bn < 0 && !mask) {
            a.sum2_out = c.dy(s2_bn);
            a.sum2_z = c.act(c.p->layers[s2_bn].producer);
            a.stats_partial = reinterpret_cast<float *>(c.base + c.ws.L[s2_bn].bwd_partial);
            a.stats_rows = &rows;
            s2 = imk_conv_can_sum2(a);
            if (!s2) { a.sum2_out = nullptr; a.sum2_z = nullptr; a.stats_partial = nullptr; a.stats_rows = nullptr; }
        }
        int rc = imk_launch_conv(a, c.stream);
        if (rc) return rc;
        if (s2) return dgrad_done(s2_bn, rows);
        return dgrad_done(stat_bn, rows);
    }
    void wgrad_args(int conv, const f16 *dA_override, ImkWgradArgs &a) const {
        const ImkLayer &l = c.p->layers[conv];
        const Dim d = res_dim(c.p->cfg, l.res);
        a.x = conv_input(c, conv);
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
        if (imk_wgrad_gemm_ok(a.x.lmode, a.dA_z != nullptr, l.ksize, a.x.cs_in, a.cs_out, (long long)c.B * d.h * d.w)) {   // wide layers: the GEMM-class kernel
            a.n_split = imk_wgrad_gemm_splits(a.x.lmode, c.B, d.h, d.w, l.ksize, a.x.cs_in, a.cs_out);
            return;
        }
        // The pooled-input 3x3 form of the kernel holds 162 VGPRs: three workgroups per CU would leave the backward chain's next
        // kernel no registers to start in (see wgrad_mfma_body: KS3), so it runs with two
        if (l.ksize == 3 && l.lmode == LM_POOL) {
            static const int pool_wgs = []() { const char *e = getenv("IMK_WGRAD_POOL_WGS"); return e ? atoi(e) : 512; }();
            const int n_pairs = ((imk_pad8(l.cin) + 15) / 16) * ((imk_pad8(l.cout) + 15) / 16);
            a.n_split = std::max(1, std::min(a.n_split, pool_wgs / n_pairs));
        }
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
    bool defer_finalize = false;
    // The end of the step: the first encoder block's 3x3 weight gradient is the heaviest of the last launches (a full-resolution
    // 3x3 from 17 channels up, where it is a launch of its own) and the side stream, not the main chain, is what finishes last.
    // Held back (hold_conv), it runs on the MAIN stream after the input block's BatchNorm backward, while the input block's
    // own, lighter weight gradient takes the side stream: -0.5 % (Cityscapes alpha 2 5.165 -> 5.142 ms, alpha 1.25 4.230 -> 4.204).
    int hold_conv = -1;
    Pending held{};
    bool has_held = false;
    // The final block of a backward pass whose last full-resolution 3x3 weight gradient is NOT held back (EvalNet's second tower): released
    // late it starts when the main chain has nothing left but the input block and is what the step then waits for (141 us alone at the
    // end of an EvalNet step); released with its dgrad it is done when the chain is (EvalNet step 2.116 -> 2.07 ms).
    bool tail_early = false;
    int launch_wgrad(int conv, const f16 *dA_override, hipStream_t ws) {
        ImkWgradArgs a{};
        wgrad_args(conv, dA_override, a);
        int rc = imk_launch_wgrad(a, ws);
        if (rc) return rc;
        return wgrad_job(conv, a);
    }
    // last = the backward pass's final weight gradient (the input block's): nothing is left to overlap with, so it stays
    // queued for finish_wgrads, which launches it on the main stream -- no fork and no join in the step's tail
    int wgrad(int conv, const f16 *dA_override = nullptr, bool last = false) {
        const ImkLayer &l = c.p->layers[conv];
        const Dim d = res_dim(c.p->cfg, l.res);
        const bool small = (long long)c.B * d.h * d.w <= side_max_pixels;
        if (n_side > 0 && small && conv == hold_conv) { held = Pending{conv, dA_override}; has_held = true; return IMK_OK; }
        if (n_side > 0 && small) {
            if (n_pending == 8) { int rc = flush_wgrads(); if (rc) return rc; }
            pending[n_pending++] = Pending{conv, dA_override};
            // full-resolution layers: fork at once -- their kernels are long (the bubble is small against them) and the
            // last block's weight gradients would otherwise all start after the main chain has ended
            return (l.res == 0 && !last) ? flush_wgrads() : IMK_OK;
        }
        return launch_wgrad(conv, dA_override, c.stream);
    }
    // End of a resolution block below full resolution: its weight gradients go to the side stream -- every block (IMK_FORK_EVERY=1,
    // the default) or every k-th one (each fork is an event record on the main stream: ~6 us before the chain's next kernel starts).
    int n_block_flush = 0;
    int flush_block() {
        static const int every = []() { const char *e = getenv("IMK_FORK_EVERY"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
        if (++n_block_flush % every != 0 && n_pending <= 4) return IMK_OK;
        return flush_wgrads();
    }
    int flush_wgrads() {
        if (n_pending == 0) return IMK_OK;
        const int si = n_fork % n_side;
        hipStream_t ws = c.p->side[si];
        // the side stream waits for the main stream's last kernel: through the event bound to that kernel when the pass runs with a
        // stop-event ring (imk_common.h: no marker packet in front of the chain's next kernel), else through a recorded event
        ImkStopRing *ring = imk_tls_stop_ring;
        if (ring && ring->stream == c.stream && ring->last) {
            IMK_HIP(hipStreamWaitEvent(ws, ring->last, 0));
            ++n_fork;
        } else {
            hipEvent_t ev = c.p->ev_fork[n_fork++];
            IMK_HIP(hipEventRecord(ev, c.stream));
            IMK_HIP(hipStreamWaitEvent(ws, ev, 0));
        }
        used_side[si] = true;
        for (int i = 0; i < n_pending; ++i) {
            int rc = launch_wgrad(pending[i].conv, pending[i].dA_override, ws);
            if (rc) return rc;
        }
        n_pending = 0;
        // ... and their split reductions right behind them, so that only the last layers' are left for the end of the step.
        // Those last layers (defer_finalize: the final block of the backward pass) skip it: their reductions would sit between
        // their weight-gradient kernels on the side stream, which the end of the step waits for; finish_wgrads does all of
        // them in one pair of launches instead.
        if (defer_finalize) return IMK_OK;
        int rc = imk_launch_wgrad_finalize_jobs(jobs, &ctl->inv_loss_scale, found_inf, ws);
        jobs = ImkWgFinalJobs{};
        return rc;
    }
    // wgrad (side stream) + dgrad of `conv`; both read its pre-activation gradient.  Running the two as ONE launch
    // (dgrad and wgrad blocks side by side in one grid) was measured for the deep layers: no gain (1.451 vs 1.452 ms per
    // step) -- the common LDS footprint of the fused kernel leaves one workgroup per CU.
    int wgrad_dgrad(int conv, f16 *dst, const f16 *mask, int stat_bn = -1) {
        const int s2_bn = sum2_bn;
        sum2_bn = -1;
        // A 1x1 conv whose forward input is the tensor its dgrad masks with (Conv3x3+ReLU -> Conv1x1): the pipelined dgrad
        // kernel has both operands of the weight gradient in LDS and produces it on the side (imk_conv_can_fuse_wgrad) --
        // no weight-gradient launch, no second read of dy, z and x.
        const ImkLayer &l = c.p->layers[conv];
        if (mask && stat_bn < 0 && l.ksize == 1 && l.lmode == LM_RAW && l.src >= 0 && mask == c.act(l.src)) {
            ImkConvArgs a{};
            int rows = 0;
            dgrad_args(conv, dst, mask, -1, nullptr, a);
            if (imk_conv_can_fuse_wgrad(a)) {
                a.wg_partial = reinterpret_cast<float *>(c.base + c.ws.L[conv].wg_partial);
                a.stats_rows = &rows;
                int rc = imk_launch_conv(a, c.stream);
                if (rc) return rc;
                if (rows <= 0 || rows > imk_conv_fused_wgrad_rows_max()) return IMK_EWORKSPACE;
                // its split reduction joins the next batch of the side stream (or the end of the step)
                return imk_wgf_add_job(jobs, a.wg_partial, rows, l.ksize, l.cin, l.cout, grads + l.off_w, grads + l.off_b);
            }
        }
        // A Conv1x1 with 24-64 channels on both sides that feeds a BatchNorm -- the blocks' second conv (its input is the ReLU mask
        // of its dgrad) and the decoder blocks' first one (upsample + add input, no mask): dgrad and weight gradient from one
        // read of dy, z and x (imk_bwd1.hip) instead of two launches that read them twice.
        if (stat_bn < 0 && l.ksize == 1 && l.bn_after >= 0 && l.src >= 0 &&
            ((l.lmode == LM_RAW && mask && mask == c.act(l.src)) || (l.lmode == LM_UPADD && !mask)) &&
            imk_bwd1x1_ok(l.lmode, imk_pad8(l.cin), imk_pad8(l.cout), mask != nullptr)) {
            const Dim d = res_dim(c.p->cfg, l.res);
            const int rows = imk_bwd1x1_rows((long long)c.B * d.h * d.w, imk_pad8(l.cin), imk_pad8(l.cout));
            float *wgp = reinterpret_cast<float *>(c.base + c.ws.L[conv].wg_partial);
            const int bn = l.bn_after;
            int rc = imk_launch_bwd1x1(conv_input(c, conv), c.dy(bn), c.act(conv), reinterpret_cast<const float *>(c.base + c.ws.L[bn].coef),
                                       c.wbwd(conv), dst, wgp, c.B, d.h, d.w, l.cout, c.stream);
            if (rc) return rc;
            return imk_wgf_add_job(jobs, wgp, rows, l.ksize, l.cin, l.cout, grads + l.off_w, grads + l.off_b);
        }
        // A 3x3 conv that reads a BatchNorm output, its dgrad feeding that BatchNorm's gradient (Conv1x1 -> BN -> Conv3x3 of
        // the decoder blocks, input block -> first encoder conv): x = BN(z) with z the tensor the gradient statistics read.
        if (!mask && stat_bn >= 0 && l.ksize == 3 && l.lmode == LM_AFFINE && l.src_bn == stat_bn &&
            l.src == c.p->layers[stat_bn].producer) {
            ImkConvArgs a{};
            int rows = 0;
            dgrad_args(conv, dst, nullptr, stat_bn, &rows, a);
            a.wg_sc = c.bn_scale(stat_bn); a.wg_sh = c.bn_shift(stat_bn);
            if (imk_conv_can_fuse_wgrad(a)) {
                a.wg_partial = reinterpret_cast<float *>(c.base + c.ws.L[conv].wg_partial);
                int rc = imk_launch_conv(a, c.stream);
                if (rc) return rc;
                rc = dgrad_done(stat_bn, rows);
                if (rc) return rc;
                if (rows > imk_conv_fused_wgrad_rows_max()) return IMK_EWORKSPACE;
                return imk_wgf_add_job(jobs, a.wg_partial, rows, l.ksize, l.cin, l.cout, grads + l.off_w, grads + l.off_b);
            }
        }
        // Full resolution: the weight gradient is released when the dgrad beside it has FINISHED, not when it starts -- both
        // are bandwidth-bound there, and the dgrad is the one the chain waits for (step -0.9 %, SUIM -1.4 %, HeLa -1.2 %; the
        // same at the lower levels changes nothing).  IMK_FORK_LATE = highest level released late (-1: none).
        // Round 5, after the input block's streaming weight gradient and the narrower staging maps: at <= 16 channels the early release
        // wins again (ISIC 0.984 -> 0.967 ms, SUIM 1.664 -> 1.647, HeLa -0.5 %, Cityscapes alpha 1 -0.9 %), from 24 channels up the late
        // one still does (Cityscapes alpha 1.5 / 2, ISIC alpha 1.5: +0.7-1 % released early) -- the rule follows the layer's width
        // unless IMK_FORK_LATE is set.
        static const bool late_set = getenv("IMK_FORK_LATE") != nullptr;
        static const int late_res = []() { const char *e = getenv("IMK_FORK_LATE"); return e ? atoi(e) : 0; }();
        const bool wide = imk_pad8(l.cin) >= 24 || imk_pad8(l.cout) >= 24;
        if (l.res <= late_res && !tail_early && (wide || late_set)) {
            int rc = dgrad(conv, dst, mask, stat_bn, s2_bn);
            if (rc) return rc;
            return wgrad(conv);
        }
        int rc = wgrad(conv);
        if (rc) return rc;
        return dgrad(conv, dst, mask, stat_bn, s2_bn);
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
            ImkStopRing *ring = imk_tls_stop_ring;
            if (ring && ring->stream == c.stream && ring->side == c.p->side[si] && ring->side_last) {
                IMK_HIP(hipStreamWaitEvent(c.stream, ring->side_last, 0));      // the side stream's last kernel's own event
                continue;
            }
            IMK_HIP(hipEventRecord(c.p->ev_join[si], c.p->side[si]));
            IMK_HIP(hipStreamWaitEvent(c.stream, c.p->ev_join[si], 0));
        }
        return imk_launch_wgrad_finalize_jobs(jobs, &ctl->inv_loss_scale, found_inf, c.stream);
    }
    // BN backward for `bn`: per-channel coefficients of  dz = A*dy + B*z + C  (+ gamma/beta gradients).  The
    // reduction (sum dy, sum dy*z) comes from the kernel that produced dy when there was exactly one (dy_rows),
    // else from a pass that also assembles dy from its sources (mode 1: skip gradient + max-pool scatter,
    // mode 2: 2x2 sum of the upsampled branch).  The consumers apply the coefficients on load (LM_BNBWD).
    int bn_bwd(int bn, int mode, const f16 *g_direct, const f16 *g_other, int g_other_cs = 0) {
        const ImkLayer &b = c.p->layers[bn];
        const Dim d = res_dim(c.p->cfg, b.res);
        const int cs = imk_pad8(b.cout);
        const int prod = b.producer;
        const LayerWs &lw = c.ws.L[bn];
        float *partial = reinterpret_cast<float *>(c.base + lw.bwd_partial);
        float *coef = reinterpret_cast<float *>(c.base + lw.coef);
        const float *save = reinterpret_cast<const float *>(c.base + lw.save);
        const f16 *z = c.act(prod);
        int rows = dy_rows[bn];
        // mode 2 with rows: the dgrad that produced the upsampled branch's gradient has assembled dy and its statistics (sum2_bn)
        if (mode == 1 || rows == 0) {
            int rc = imk_launch_bn_bwd_prep(mode, mode == 0 ? c.dy(bn) : g_direct, g_other, z, c.bn_scale(bn), c.bn_shift(bn),
                                            c.dy(bn), partial, c.B, d.h, d.w, cs, c.stream, g_other_cs);
            if (rc) return rc;
            rows = imk_bn_prep_blocks(c.B, d.h, d.w, cs);
        }
        return imk_launch_bn_bwd_coef(partial, rows, b.cout, cs, (double)c.B * d.h * d.w, c.params + b.off_w, save, save + cs,
                                      &ctl->inv_loss_scale, coef, grads + b.off_w, grads + b.off_b, found_inf, c.stream);
    }
};

// Installs the plan's stop-event ring for `stream` on this thread for the lifetime of the object (a training step's backward pass).
// IMK_STOP_EVENTS=0: forks record events of their own, as rounds 2-4 did.
struct ImkStopRingScope {
    ImkStopRing ring{};
    ImkStopRing *prev = nullptr;
    bool on = false;
    ImkStopRingScope(const imk_unet_plan *plan, hipStream_t stream, int n_side) {
        const bool want = n_side > 0;
        // IMK_STOP_EVENTS: 0 = off, 1 (default) = the forks wait for kernel-bound events, 2 = the join at the end of the step as well
        // (one box, four repetitions: ISIC step 0.964 / 0.954 / 0.957 ms, SUIM 1.631 / 1.614 / 1.620, EvalNet 2.037 / 2.027 / 2.025)
        static const int mode = []() { const char *e = getenv("IMK_STOP_EVENTS"); return e ? atoi(e) : 1; }();
        if (!want || mode <= 0 || !plan->ev_ring[0]) return;
        ring = ImkStopRing{plan->ev_ring, 128, 0, stream, nullptr};
        if (n_side == 1 && mode >= 2) ring.side = plan->side[0];
        prev = imk_tls_stop_ring;
        imk_tls_stop_ring = &ring;
        on = true;
    }
    ~ImkStopRingScope() { if (on) imk_tls_stop_ring = prev; }
    ImkStopRingScope(const ImkStopRingScope &) = delete;
    ImkStopRingScope &operator=(const ImkStopRingScope &) = delete;
};

// Ensemble inference + IM.  Workspace: [N][B,H,W,K] fp32 probabilities, then one model's activations.
// Side streams (training: weight gradients; ensemble inference: one model per stream) are shared by every plan of the
// process on a device; the fork / join events are the plan's own.  One pair per PLAN cost 15 % of a SUIM training step
// in bench.py: three teachers + the student = 8 streams on the runtime's 4 hardware queues, the student's side stream
// landed on the queue of its main stream and the "concurrent" weight gradients ran in line behind barrier packets
// (1.99 ms per step; 1.73 with GPU_MAX_HW_QUEUES=8 or with this pool).  Streams created here live as long as the process.
// Only as many as a call needs are created (training: one; an ensemble of three: two): every stream beyond the hardware queues
// costs concurrency somewhere -- with RCCL's own stream in the process, a second idle pool stream was enough to put the weight
// gradients in line behind the main chain again (forced one-rank run: 20.6 k instead of 23.7 k images/s; package __init__
// also raises GPU_MAX_HW_QUEUES to 8 when it is imported before the HIP runtime starts).
// The pool itself lives in ONE translation unit (imk_unet.hip: imk_side_pool_stream, declared in imk_kernels.h): this header is
// included by imk_unet.hip and imk_evalnet.hip, and a pool per translation unit gave an IM++ process two sets of streams.
// the plan's events (once) and its first n side streams
inline bool ensure_side_streams(const imk_unet_plan *plan, int n = 1) {
    std::call_once(plan->side_once, [plan]() {
        bool ok = true;
        for (int i = 0; i < imk_unet_plan::MAX_SIDE; ++i)
            ok = ok && hipEventCreateWithFlags(&plan->ev_join[i], hipEventDisableTiming) == hipSuccess;
        for (auto &e : plan->ev_fork) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto &e : plan->ev_ring) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        plan->side_ok = ok;
    });
    if (!plan->side_ok) return false;
    for (int i = 0; i < n && i < imk_unet_plan::MAX_SIDE; ++i) {
        if (!plan->side[i]) plan->side[i] = imk_side_pool_stream(i);
        if (!plan->side[i]) return false;
    }
    return true;
}

// ctl / stats: non-null after an optimizer step -- the first packing launch then also closes the step (loss-scale and
// step-counter update), which saves a launch of its own.
inline int pack_weights(const imk_unet_plan *plan, const float *params, void *packed, hipStream_t stream, ImkCtl *ctl,
                        const float *stats, bool fold_bn) {
    IMK_CHECK_ARG(plan && params && packed);
    uint8_t *pk = (uint8_t *)packed;
    ImkPackJobs pj{};
    ImkFoldJobs fj{};
    pj.ctl = ctl; pj.stats = stats;
    auto flush_pack = [&]() -> int { int rc = imk_launch_pack_jobs(pj, stream); pj.n = 0; pj.ctl = nullptr; return rc; };
    for (size_t i = 0; i < plan->layers.size(); ++i) {
        const ImkLayer &l = plan->layers[i];
        if (l.kind == 0) {
            if (l.flags & IMK_LF_DENSE) continue;
            for (int tr = 0; tr < 3; ++tr) {
                if (tr == 0 && (l.flags & IMK_LF_HEAD)) continue;   // the head runs in fp32 from `params` directly
                if (tr == 2 && (l.pk_chain < 0 || (l.flags & IMK_LF_HEAD))) continue;
                f16 *dst = (f16 *)(pk + (tr == 2 ? l.pk_chain : (tr ? l.pk_bwd : l.pk_fwd)));
                const int pair = tr == 2 ? (imk_conv_pair_layout(l.cin, l.cout, false) && l.cin <= 8)
                                         : (tr ? imk_conv_pair_layout(l.cout, l.cin, false) : imk_conv_pair_layout(l.cin, l.cout, l.lmode == LM_U8));
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

}  // namespace
