// Host-side description of a network built from the reference's blocks -- the tiny U-Net (unet.py:4-67) and EvalNet
// (evalnet.py:4-47): a list of conv / BatchNorm layers plus, per layer, where its input comes from.  Shared by the
// forward / training code of both (imk_net.h).
#pragma once
#include <mutex>
#include <string>
#include <vector>
#include "imk_common.h"

struct ImkLayer {
    std::string name;
    int kind;      // 0 conv, 1 bn
    int ksize, cin, cout;
    int64_t off_w, off_b;        // conv: kernel/bias; bn: gamma/beta (floats, trainable section)
    int64_t off_mean, off_var;   // bn (non-trainable section)
    int res;       // resolution level of the layer's output: 0 = full ... 4 = 1/16
    // packed (fp16, MFMA fragment order) weights, byte offsets into the packed buffer
    int64_t pk_fwd, pk_bwd;      // conv only
    int64_t pk_bytes_fwd, pk_bytes_bwd;
    int64_t pk_chain;            // 1x1 convs: chain operand (fused behind the preceding 3x3), else -1
    int64_t pk_scale;            // bn: fp32 scale[cpad], shift[cpad] for inference (folded moving stats)
    // graph, filled by the network builder
    int src = -1;                // conv: the conv whose output it reads, or IMK_SRC_XA / _XB (the uint8 network inputs) /
                                 // IMK_SRC_CAT (EvalNet's concatenated towers) / IMK_SRC_ONEHOT (its one-hot expanded input B)
    int src_bn = -1;             // conv: BatchNorm applied to `src` on load (-1: none)
    int src2 = -1, src2_bn = -1; // conv, LM_UPADD: the skip tensor and its BatchNorm
    int lmode = 0;               // conv: ImkLoadMode
    int bn_after = -1;           // conv: the BatchNorm that normalises its output (-1: none)
    int producer = -1;           // bn: the conv that feeds it
    int flags = 0;               // IMK_LF_*
};
enum { IMK_SRC_XA = -2, IMK_SRC_XB = -3, IMK_SRC_CAT = -4, IMK_SRC_ONEHOT = -5 };
enum {
    IMK_LF_HEAD = 1,     // the U-Net's output conv: forward in fp32 from the flat parameters (only its dgrad operand is packed)
    IMK_LF_DENSE = 4,    // EvalNet's Dense heads: fp32 only, no packed weights, no workspace
    IMK_LF_U8_RAW = 2,   // uint8 stem without the x/255 Lambda (evalnet.py:5-6, normalize=False)
};

struct imk_unet_plan {
    imk_unet_cfg cfg;            // EvalNet plans: h, w, c_in (= input A's channels), n_out and ch are filled too
    int net = 0;                 // 0 U-Net, 1 EvalNet
    imk_evalnet_cfg ecfg{};
    std::vector<ImkLayer> layers;
    int64_t n_total, n_trainable;
    int64_t packed_bytes;
    // Side stream for the weight-gradient kernels (off the dependency chain of the backward pass), forked
    // from / joined to the caller's stream with events (one fork per conv layer: 24).  Created on the first training call.
    mutable std::once_flag side_once;
    static constexpr int MAX_SIDE = 2;
    mutable hipStream_t side[MAX_SIDE] = {};
    mutable hipEvent_t ev_fork[40] = {};
    mutable hipEvent_t ev_join[MAX_SIDE] = {};
    mutable hipEvent_t ev_ring[128] = {};     // kernel-bound stop events of the backward pass (ImkStopRing, imk_common.h)
    mutable bool side_ok = false;
    // Debug / measurement switches of THIS plan (imk_unet_plan_debug; the caller owns the plan, the library keeps no global):
    // materialize: inference also stores the intermediates of fused kernels (layer-by-layer parity);
    // single_stream: no side streams -- every kernel alone on the caller's stream (exclusive kernel timings).
    bool dbg_materialize = false, dbg_single_stream = false;
    // momentum of the BatchNorm moving statistics in training steps (imk_unet_plan_set_bn_momentum; Keras default 0.99)
    float bn_momentum = 0.99f;
    int find(const char *name) const {
        for (size_t i = 0; i < layers.size(); ++i) if (layers[i].name == name) return (int)i;
        return -1;
    }
};
