// Internal launch API between the network orchestration (imk_net.h, imk_unet.hip, imk_evalnet.hip) and the conv kernels.
#pragma once
#include "imk_common.h"

// How a conv kernel materialises its input tile (the producer's BatchNorm is applied on load).
enum ImkLoadMode {
    LM_RAW = 0,     // fp16 [B,H,W,cs] as is
    LM_AFFINE = 1,  // fp16(z*sc + sh)                                   (BN on load)
    LM_POOL = 2,    // 2x2 max of fp16(z*sc + sh), z at [B,src_h,src_w,cs] (BN + MaxPooling2D, unet.py:16-17)
    LM_UPADD = 3,   // fp16( fp16(zlo*sc+sh)[y/2,x/2] + fp16(zsk*sc2+sh2) )  (UpSampling2D + add, unet.py:32-33)
    LM_U8 = 4,      // fp16(u8/u8_div), [B,H,W,cin] bytes                (Lambda x/255, unet.py:5)
    LM_BNBWD = 5,   // (A*dy + B*z + C) * [z > 0]: BatchNorm backward + ReLU backward applied on load;
                    // in = dy, in2 = z (the BN's input), sc = per-channel coefficients [A | B | C] (3*cs floats)
    LM_STEM = 6,    // inference: the whole input block on load (unet.py:4-9: x/255 -> Conv1x1+ReLU -> BatchNorm):
                    // in = uint8 image [B,H,W,u8_c], sc2 / sh2 = the stem's fp32 kernel [u8_c][cin] / bias [cin],
                    // sc / sh = its folded BatchNorm; fp16(fp16(relu(W . fp16(x/255) + b)) * sc + sh), i.e. what
                    // LM_AFFINE reads from the stored stem output.  Pipelined kernel only (<= 16 channels).
};
enum ImkEpilogue {
    EP_RELU = 0,   // fp16(max(acc + bias, 0)); optional per-channel sum / sum-of-squares partials
    EP_PLAIN = 1,  // fp16(acc)                      (dgrad)
    EP_MASK = 2,   // fp16(mask > 0 ? acc : 0)       (dgrad through the ReLU of the producing conv)
};

struct ImkInput {
    const void *in;        // see ImkLoadMode
    const void *in2;       // LM_UPADD: skip tensor
    const float *sc, *sh;  // [cs_in] affine of `in`
    const float *sc2, *sh2;
    int lmode;
    int cin, cs_in;        // logical / padded-to-8 channel count (LM_U8: cs_in = 8)
    float u8_div;          // LM_U8: 255 (Lambda x/255, unet.py:5) or 1 (evalnet.py:5, normalize=False)
    int u8_c;              // LM_STEM: bytes per pixel of the uint8 image (<= 4)
    int src_h, src_w;      // LM_POOL: size of the tensor that is pooled (0: 2H x 2W); 2H + 1 / 2W + 1 when MaxPooling2D dropped
                           // an odd last row / column (Keras 'valid' pooling: EvalNet at sizes that are not multiples of 64)
};

// tile counters of a launch that walks its tiles dynamically (ImkConvArgs::sched; imk_stage.h: ImkWalk): 32 counters, 256 B apart
#define IMK_SCHED_HEADS 32
#define IMK_SCHED_STRIDE 256
#define IMK_SCHED_BYTES (IMK_SCHED_HEADS * IMK_SCHED_STRIDE)
struct ImkConvArgs {
    ImkInput x;
    int B, H, W;           // resolution of the conv (= of the output)
    int ksize;             // 1 or 3
    int cout, cs_out;
    const f16 *wpk;        // packed weights, fragment order (see pack_conv_weights)
    const float *bias;     // [cout], EP_RELU only
    f16 *out;              // [B,H,W,cs_out]
    const f16 *mask;       // EP_MASK: [B,H,W,cs_out]
    // optional fused second stage (EP_RELU only): out2 = relu(W2 . relu(W . x + bias) + bias2), a 1x1 conv chained on
    // the accumulator tile inside the same kernel (unet.py:12-13 / 37-38: Conv3x3+ReLU -> Conv1x1+ReLU).  `out` (the
    // intermediate) is then written only if non-null; statistics are taken on out2.
    const f16 *wpk2;       // chain-packed 1x1 weights (pack mode 2)
    const float *bias2;
    f16 *out2;             // [B,H,W,cs_out2]
    int cout2, cs_out2;
    const f16 *dystat_z;   // EP_PLAIN / EP_MASK, optional: the output is a BatchNorm's output gradient dy; accumulate
                           // (sum dy, sum dy*z) with z = this tensor [B,H,W,cs_out] into stats_partial
    float *stats_partial;  // optional: [rows][2*cs] (EP_RELU: sum, sumsq of the fp16-rounded outputs)
    int *stats_rows;       // host, optional: receives the number of partial rows this launch writes
    int epi;
    int pair;              // set by imk_launch_conv: weights are in the pair layout (imk_conv_pair_layout)
    // optional, dgrad of a 1x1 conv whose forward input is exactly `mask` (Conv3x3+ReLU -> Conv1x1: the ReLU mask of the
    // dgrad IS the conv's input x): the same launch also produces the conv's weight / bias gradient partials
    // dW[ci][co] = sum_px x[px][ci] * dA[px][co] -- both operands are in LDS anyway -- one row [2][256] per workgroup
    // (stats_rows receives the row count); see imk_conv_can_fuse_wgrad.
    // optional fused FIRST stage (inference, conv_pipe_kernel<..., PRE>): the conv's input is fp16(relu(Wp . x + pre_bias)) passed
    // through a BatchNorm (pre_sc, pre_sh; zero padding outside the image), Wp a 1x1 conv with at most 16 input channels on the
    // tensor `x` describes (uint8 image or upsample + skip).  pre_wpk = that conv's regular forward pack (same pair / plain
    // layout as this conv's), pre_cout = its output channels = this conv's input channels.  See imk_conv_can_prestage.
    const f16 *pre_wpk;
    const float *pre_bias, *pre_sc, *pre_sh;
    int pre_cout;
    float *wg_partial;
    // second form (the U-Net's output layer: 1x1 conv on a BatchNorm output, dgrad = LM_RAW / EP_PLAIN with the BN-gradient
    // statistics): x = fp16(dystat_z * wg_sc + wg_sh), the BatchNorm being applied to the transposed LDS reads
    const float *wg_sc, *wg_sh;
    // optional, dgrad of a decoder block's Conv1x1 on upsample + skip (LM_BNBWD, EP_PLAIN, full tiles): the output is the gradient of
    // the upsample + add tensor, whose 2x2 sums are the gradient of the lower block's BatchNorm output (UpSampling2D's backward,
    // unet.py:33).  The launch then also writes those sums to sum2_out [B,H/2,W/2,cs_out] and their BatchNorm-backward statistics
    // against sum2_z (that BatchNorm's input, same shape) into stats_partial / stats_rows -- the "assemble dy" pass of that
    // BatchNorm (bn_bwd_prep_kernel<2>) disappears.  See imk_conv_can_sum2.
    f16 *sum2_out;
    const f16 *sum2_z;
    // optional (forward launches without statistics, i.e. inference): IMK_SCHED_BYTES of zeroed tile counters, one per group of the
    // persistent walk -- the workgroups then TAKE their tiles (imk_stage.h: ImkWalk, dynamic form) instead of striding over them
    unsigned *sched;
};
// multiply-adds x 2 of a conv launch (logical channel counts; + the chained 1x1, + the first-stage 1x1)
inline double imk_conv_flops(const ImkConvArgs &a) {
    const double px = (double)a.B * a.H * a.W;
    const int cin_main = a.pre_wpk ? a.pre_cout : a.x.cin;
    double f = 2.0 * px * (a.ksize == 3 ? 9 : 1) * cin_main * a.cout;
    if (a.wpk2) f += 2.0 * px * a.cout * a.cout2;
    if (a.pre_wpk) f += 2.0 * px * a.x.cin * a.pre_cout;
    return f;
}
bool imk_conv_can_fuse_wgrad(const ImkConvArgs &dgrad_args);
bool imk_conv_can_sum2(const ImkConvArgs &dgrad_args);        // this dgrad launch can also emit the 2x2 sums of its output (ImkConvArgs::sum2_out)
// GEMM-class kernel for the wide layers (imk_gemm.hip): which launches it takes, its launcher, its statistics rows
bool imk_conv_gemm_ok(const ImkConvArgs &a);
int imk_launch_conv_gemm(const ImkConvArgs &a, hipStream_t stream);
bool imk_conv_gemm_chain_ok(const ImkConvArgs &a);      // Conv3x3 -> Conv1x1 in one GEMM-class launch (inference)
int imk_conv_gemm_num_tiles(int B, int H, int W);
// algorithmic bytes of a conv launch: its input tensor(s) as the load mode reads them + every tensor it writes / re-reads
inline double imk_conv_algorithmic_bytes(const ImkConvArgs &a) {
    const double px = (double)a.B * a.H * a.W;
    double in_b;
    switch (a.x.lmode) {
        case LM_POOL: in_b = 4.0 * px * a.x.cs_in * 2; break;                     // reads the 2H x 2W tensor
        case LM_UPADD: in_b = px * a.x.cs_in * 2 + 0.25 * px * a.x.cs_in * 2; break;  // skip + low-res tensor
        case LM_BNBWD: in_b = 2.0 * px * a.x.cs_in * 2; break;                        // dy and z
        case LM_U8: in_b = px * a.x.cin; break;
        case LM_STEM: in_b = px * a.x.u8_c; break;
        default: in_b = px * a.x.cs_in * 2;
    }
    double out_b = (a.wpk2 && !a.out) ? 0.0 : px * a.cs_out * 2;
    if (a.wpk2) out_b += px * a.cs_out2 * 2;
    if (a.epi == EP_MASK) out_b += px * a.cs_out * 2;
    if (a.epi != EP_RELU && a.dystat_z) out_b += px * a.cs_out * 2;
    return in_b + out_b;
}
int imk_conv_fused_wgrad_rows_max();   // capacity the partial buffer needs: rows (workgroups) of the largest such launch
int imk_conv_num_tiles(int B, int H, int W, int cs_in, int ksize);  // rows of stats_partial
int imk_launch_conv(const ImkConvArgs &a, hipStream_t stream);

struct ImkWgradArgs {
    ImkInput x;            // the conv's input, same modes as forward
    const f16 *dA;         // [B,H,W,cs_out] gradient w.r.t. the conv's pre-activation output (loss-scaled) ...
    const f16 *dA_z;       // ... or, if non-null: dA is the following BatchNorm's output gradient dy, dA_z its input z,
    const float *dA_coef;  //     and the pre-activation gradient (A*dy + B*z + C)*[z > 0] is formed on load
    int B, H, W, ksize, cout, cs_out;
    float *partial;        // [n_split][n_pairs][taps+1][256] fp32 scratch
    int n_split;
};
inline double imk_wgrad_flops(const ImkWgradArgs &a) {   // 2 * pixels * taps * cin * cout (logical channels), as the forward conv
    return 2.0 * a.B * a.H * a.W * (a.ksize == 3 ? 9 : 1) * (double)a.x.cin * a.cout;
}
// Conv1x1 (+ ReLU, + BatchNorm behind it) with 24-64 channels on both sides: dgrad and weight gradient in one kernel (imk_bwd1.hip)
bool imk_bwd1x1_ok(int lmode, int cs_in, int cs_out, bool masked);
int imk_bwd1x1_rows(long long n_pix, int cs_in, int cs_out);       // workgroups = weight-gradient partial rows
int imk_launch_bwd1x1(const ImkInput &x, const f16 *dy, const f16 *z, const float *coef, const f16 *wpk_bwd, f16 *dx,
                      float *wg_partial, int B, int H, int W, int cout, hipStream_t stream);
int imk_wgrad_splits(int B, int H, int W, int cin, int cout);
// GEMM-class weight-gradient kernel of the wide layers (imk_gemm.hip).  bnb: the gradient operand is a BatchNorm backward on load
bool imk_wgrad_gemm_ok(int lmode, bool bnb, int ksize, int cs_in, int cs_out, long long pixels);
bool imk_wgrad_gemm_wide(int cs_in, int cs_out, long long pixels);   // the size rule alone (workspace sizing)
int imk_wgrad_gemm_splits(int lmode, int B, int H, int W, int ksize, int cs_in, int cs_out);   // = ImkWgradArgs::n_split of such a launch
int imk_launch_wgrad_gemm(const ImkWgradArgs &a, hipStream_t stream);
size_t imk_wgrad_partial_floats(int B, int H, int W, int ksize, int cin, int cout);
int imk_launch_wgrad(const ImkWgradArgs &a, hipStream_t stream);

// weight packing (fp32 HWIO -> fp16 fragment order).  transposed = 1 gives the dgrad operand; transposed = 2 the
// "chain" operand of a 1x1 conv applied to an accumulator tile (k-slot (g, j<4) <-> input channel 4g + j).
inline long long imk_conv_max_pixels() { return 1ll << 24; }   // H * W of a plan (the shallow kernel's 24-bit offset arithmetic)
bool imk_conv_can_chain(const ImkConvArgs &first, int cout2);
// can the 1x1 conv (cin_pre -> cout_pre channels, input mode lm_pre) run as the first stage of this (chained) 3x3 launch?
bool imk_conv_can_prestage(const ImkConvArgs &main, int lm_pre, int cin_pre, int cout_pre);
// ... by the per-tile kernel (17-64 channels), which takes the 1x1's regular forward pack as ImkConvArgs::wpk2
bool imk_conv_can_chain_tile(const ImkConvArgs &first, int cout2, bool store_intermediate);
// inference: can the input block (u8_c image channels -> ch0) be computed on load by the conv that follows it (LM_STEM)?
bool imk_conv_stem_fusable(int u8_c, int ch0, int cout_next);
// "Pair" fragment layout (see conv_pipe_kernel): used by every conv operand with <= 8 output channels that the
// pipelined kernel runs (<= 16 input channels, u8 input with <= 4 channels).  k_in / m_out are the operand's own input /
// output channel counts (forward: cin / cout, dgrad: cout / cin; chain: both must be <= 8).
bool imk_conv_pair_layout(int k_in, int m_out, bool u8_input);
size_t imk_packed_conv_halfs(int ksize, int cin, int cout, int transposed, bool pair);

// batched variant: up to IMK_PACK_MAX_JOBS (layer, direction) jobs per launch, table passed by value
#define IMK_PACK_MAX_JOBS 80
struct ImkPackJob { const float *w; f16 *dst; int ksize, cin, cout, transposed, pair; };
struct ImkCtl;
struct ImkPackJobs {
    ImkPackJob j[IMK_PACK_MAX_JOBS];
    int n;
    ImkCtl *ctl;            // optional: end-of-optimizer-step bookkeeping done by one thread of this launch
    const float *stats;
};
int imk_launch_pack_jobs(const ImkPackJobs &jobs, hipStream_t stream);

// All weight-gradient reductions of a training step in two launches (stage 1: 16 splits -> 1 chunk, coalesced;
// stage 2: <= 64 chunks -> 1, scale by 1/loss_scale, non-finite check, scatter into the flat gradient vector).
#define IMK_WGF_MAX_JOBS 32
struct ImkWgFinalJob {
    const float *partial;   // [n_split][n_tiles][256]
    float *red;             // [n_chunks][n_tiles][256]
    float *dw, *db;
    int n_split, n_chunks, chunk, n_tiles, T, cin, cout, cot_n;   // chunk: splits summed per stage-1 work item
    int work1_begin;        // prefix sum of n_tiles * n_chunks
    int tile_begin;         // prefix sum of n_tiles
};
struct ImkWgFinalJobs { ImkWgFinalJob j[IMK_WGF_MAX_JOBS]; int n, total_work1, total_tiles; };
int imk_wgf_add_job(ImkWgFinalJobs &jobs, float *partial, int n_split, int ksize, int cin, int cout, float *dw, float *db);
int imk_launch_wgrad_finalize_jobs(const ImkWgFinalJobs &jobs, const float *inv_scale_ptr, float *found_inf, hipStream_t stream);

// Side streams (training: weight gradients; ensemble inference: one model per stream): ONE pool per device for the whole
// process, defined in imk_unet.hip.  Stream i of the calling thread's current device, created on first use; nullptr on failure.
// The first creation also checks GPU_MAX_HW_QUEUES (imk_runtime_warnings, include/imk.h).
hipStream_t imk_side_pool_stream(int i);
