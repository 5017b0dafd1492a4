// Host-side description of the tiny U-Net (unet.py:4-67) shared by the forward / training code.
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
};

struct imk_unet_plan {
    imk_unet_cfg cfg;
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
    mutable bool side_ok = false;
    int find(const char *name) const {
        for (size_t i = 0; i < layers.size(); ++i) if (layers[i].name == name) return (int)i;
        return -1;
    }
};
