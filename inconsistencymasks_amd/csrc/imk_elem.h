// Launchers of the streaming kernels in imk_elem.hip.
#pragma once
#include "imk_common.h"

// Device-resident control block of the training state (no host round trips inside a step).
struct ImkCtl {
    float loss_scale;      // dynamic loss scale (Keras LossScaleOptimizer semantics)
    float inv_loss_scale;
    int good_steps;        // consecutive finite steps since the last scale change
    int step;              // number of applied optimizer steps
    float found_inf;       // unused (the overflow flag of a step lives in its stats[1], where the all-reduce sees it)
    float pad[3];
};

int imk_launch_bn_finalize(const float *partial, int n_part, int c, int cs, double count, const float *gamma,
                           const float *beta, float *mov_mean, float *mov_var, float *scale, float *shift,
                           float *save_mean, float *save_invstd, hipStream_t stream, float momentum = 0.99f);   // Keras default momentum
int imk_bn_prep_blocks(int B, int H, int W, int cs);
int imk_launch_bn_bwd_prep(int mode, const f16 *g_direct, const f16 *g_other, const f16 *z, const float *sc,
                           const float *sh, f16 *dy_out, float *partial, int B, int H, int W, int cs, hipStream_t stream,
                           int g_other_cs = 0);   // mode 1 without g_direct: channel stride of g_other (0: cs)
int imk_launch_bn_bwd_coef(const float *partial, int n_part, int c, int cs, double count, const float *gamma,
                           const float *save_mean, const float *save_invstd, const float *inv_scale_ptr, float *coef,
                           float *dgamma, float *dbeta, float *found_inf, hipStream_t stream);
int imk_launch_head(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                    int K, int softmax, long long n_pix, float *probs, hipStream_t stream);
int imk_loss_blocks(long long n_pix);
// training: head (BN on load, fp32 1x1 conv, sigmoid / softmax) + loss + d(loss * scale)/d(logits), no probability tensor
int imk_launch_head_loss(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                         int K, int softmax, long long n_pix, const uint8_t *y, const ImkCtl *ctl, float *stats,
                         f16 *dlogit, float *loss_partial, hipStream_t stream);
// softmax heads, training: head + loss + gradient of the last BatchNorm's output (+ its statistics) + the output layer's
// weight / bias gradient partials in ONE pass (imk_headf.hip); rows = workgroups = rows of every partial buffer it writes
bool imk_head_cce_fused_ok(int cs, int K, long long n_pix, int rows_cap);
int imk_head_cce_fused_rows(long long n_pix);
int imk_launch_head_cce_fused(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                              int K, long long n_pix, const uint8_t *y, const ImkCtl *ctl, float *stats, f16 *dy,
                              float *loss_partial, float *dystat_partial, float *wg_partial, hipStream_t stream);
// the same for sigmoid heads (mse) with 1 or 3 maps and <= 16 input channels (imk_headf.hip: head_mse_fused_kernel)
bool imk_head_mse_fused_ok(int cs, int K, long long n_pix, int rows_cap);
int imk_launch_head_mse_fused(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                              int K, long long n_pix, const uint8_t *y, const ImkCtl *ctl, float *stats, f16 *dy,
                              float *loss_partial, float *dystat_partial, float *wg_partial, hipStream_t stream);
int imk_launch_loss_finalize(const float *loss_partial, long long n_pix, int K, int kind, float *stats, hipStream_t stream);
int imk_launch_ctl_init(ImkCtl *ctl, hipStream_t stream);
int imk_launch_adamw(float *p, float *m, float *v, const float *g, long long n, ImkCtl *ctl, const float *stats,
                     float grad_scale, float lr, float wd, float b1, float b2, float eps, hipStream_t stream);

// EvalNet (evalnet.py:24-47): concatenation of the towers' pooled BatchNorm outputs; the tail (BN + pool on load, global
// average pool, Dense + sigmoid head(s), and in training the losses, the gradient of the pooled map and the per-sample
// Dense gradients) and the batch reduction of the latter
int imk_launch_concat_pool(const f16 *za, const float *sca, const float *sha, int csa, const f16 *zb, const float *scb,
                           const float *shb, int csb, int B, int Hh, int Wh, f16 *cat, hipStream_t stream);
int imk_launch_onehot(const uint8_t *cls, long long n_pix, int cs, f16 *out, hipStream_t stream);
size_t imk_evalnet_head_partial_floats(int B, int n_heads, int K, int C);
int imk_launch_evalnet_head(const f16 *z, const float *sc, const float *sh, const float *const *w, const float *const *bias,
                            int n_heads, int K, int C, int cs, int B, int H, int W, float *out, const float *y,
                            const ImkCtl *ctl, float *stats, f16 *dP, float *partial, hipStream_t stream);
int imk_launch_evalnet_head_reduce(const float *partial, int B, int n_heads, int K, int C, const float *inv_scale_ptr,
                                   float *dw0, float *db0, float *dw1, float *db1, float *found_inf, float *stats,
                                   hipStream_t stream);

// End of an optimizer step (Keras dynamic loss scale: halve on overflow, double after 2000 consecutive finite steps;
// step counter).  Called by one thread of the weight re-packing kernel, which runs right after adamw_kernel.
__device__ __forceinline__ void imk_ctl_end_step(ImkCtl *ctl, const float *stats) {
    if (stats[1] != 0.f) {
        ctl->loss_scale = fmaxf(ctl->loss_scale * 0.5f, 1.0f);
        ctl->good_steps = 0;
    } else {
        ctl->step += 1;
        if (++ctl->good_steps >= 2000) { ctl->loss_scale *= 2.0f; ctl->good_steps = 0; }
    }
    ctl->inv_loss_scale = 1.0f / ctl->loss_scale;
}

// all BatchNorm layers of a model folded (moving statistics -> scale | shift) in one launch
#define IMK_FOLD_MAX_JOBS 32
struct ImkFoldJob { const float *gamma, *beta, *mean, *var; float *scale; int c, cs; };
struct ImkFoldJobs { ImkFoldJob j[IMK_FOLD_MAX_JOBS]; int n; };
int imk_launch_bn_fold_jobs(const ImkFoldJobs &jobs, hipStream_t stream);
