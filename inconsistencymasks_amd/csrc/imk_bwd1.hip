// Backward pass of a Conv1x1 (+ ReLU, followed by a BatchNorm) with 24-64 channels on both sides in ONE kernel: the gradient
// w.r.t. its input AND its weight / bias gradient partials, from one read of the three tensors both need.
//
// The blocks of unet.py:11-19, 32-41 (and evalnet.py:8-15) end in Conv1x1 -> ReLU -> BatchNorm.  Backward, such a conv needs
//   dA = (A dy + B z + C) [z > 0]            the BatchNorm-backward + ReLU-backward of its output (dy, z: [pixels, cout])
//   dX = (dA . W^T) [x > 0]                  (x = the Conv3x3+ReLU output it read: unet.py:12-13 -- also the ReLU mask of dX)
//   dW = x^T . dA,  db = column sums of dA
// i.e. two skinny GEMMs over the same three tensors.  As separate dgrad and weight-gradient launches (conv_gemm_kernel /
// conv_wide_kernel + wgrad_gemm_kernel / wgrad_mfma_kernel) they read dy, z and x twice: 7 tensor passes for an operation that
// is bandwidth-bound at these widths (1x1, <= 64 channels: ~60 FLOP/B).  Here: 3 reads + 1 write.  At Cityscapes alpha = 2
// the 1x1 weight gradients alone were 19 % of a training step (profiles/README.md, round 3).
//
//   * 128 consecutive pixels of the flattened [B*H*W] tensor per iteration (a 1x1 conv has no halo: no 2-D tiling, every load
//     and store a full line), persistent workgroups, the next tile's loads in flight in registers during the MFMAs;
//   * dA and x are staged ONCE as per-16-channel slices [pixel][16] (32-byte pitch): the dgrad reads a pixel's 8-channel chunk
//     with one ds_read_b128 (B operand, contraction over cout), the weight gradient reads the same slices transposed
//     (ds_read_tr16_b64, contraction over pixels) -- one LDS image, both products;
//   * the upsample + add form (decoder blocks' first conv, unet.py:32-35: x = up(BN(lo)) + BN(skip), no ReLU mask on dX) is the
//     same kernel with LM_UPADD staging for x;
//   * a wave owns its weight-gradient accumulators (one input-channel tile x all output tiles): no cross-wave reduction, one
//     partial row per workgroup in wgf_stage1's layout; fixed orders everywhere (bit-reproducible).
#include <cstdlib>
#include "imk_stage.h"

IMK_STAMP_TABLE(bwd1)

namespace {

struct Bwd1Args {
    ImkInput x;               // the conv's forward input: LM_RAW (= the dgrad's ReLU mask) or LM_UPADD
    const f16 *dy, *z;        // [pixels][cs_o]: the following BatchNorm's output gradient and input
    const float *coef;        // [3][cs_o] A | B | C of that BatchNorm's backward
    const f16 *wpk;           // dgrad operand of the conv (pack mode 1: rows = input channels, k = output channels)
    f16 *dx;                  // [pixels][cs_i]
    float *wg_partial;        // [grid][cit_n * cot_n][2][256]
    int B, H, W, cin, cs_i, cout, cs_o;
    int nc8o, nc8p, n_pass;   // k geometry of the dgrad pack: output-channel chunks, chunks per pass, passes (<= 2)
    int cit_n, cot_n;
    long long n_pix;
};

// SMALL: at most 32 channels on both sides (2 x 2 tiles, one k-step) -- the full-resolution layers, where the kernel's time
// is: 4 instead of 2 weight-gradient partial sums per input-channel tile (every wave works: wave = (tile, half of the
// tile's 4 k-steps), combined in a fixed order at the end), a quarter of the fragment / accumulator registers.
#ifndef BWD1_SMALL_WGS
#define BWD1_SMALL_WGS 3
#endif
template <int LM, bool SMALL>
__global__ __launch_bounds__(256, SMALL ? BWD1_SMALL_WGS : 2) void bwd1x1_kernel(Bwd1Args a) {
    IMK_STAMP_BEGIN(bwd1, 70000 + LM * 10 + (SMALL ? 1 : 0));
    constexpr int NPX = 128, H16 = WG_STRIDE_H;
    constexpr int NF = SMALL ? 2 : 4, NSK = SMALL ? 1 : 2;   // input-channel tiles (= output tiles) and dgrad k-steps at most
    constexpr bool MASK = LM == LM_RAW;
    // staging slots per thread and tensor: 128 pixels x CH 8-channel chunks / 256 threads.  SMALL rows have at most 4 chunks: a slot
    // per chunk that exists (the 8-chunk map left half of the threads loading a clamped duplicate and storing zeros to unused slices)
    constexpr int CH = SMALL ? 4 : 8, CSH = SMALL ? 2 : 3, NS = NPX * CH / 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NSL = SMALL ? 2 : 4;                      // 16-channel slices per tensor
    f16 *s_d = reinterpret_cast<f16 *>(smem);               // [NSL][128][16]  dA slices
    f16 *s_x = s_d + NSL * NPX * H16;                       // [NSL][128][16]  x slices
    f16 *s_o = s_x + NSL * NPX * H16;                       // [128][72]       dX tile for the coalesced copy-out
    float *s_coef = reinterpret_cast<float *>(s_o + NPX * 72);   // [3][cs_o]
    float *s_aff = s_coef + 3 * a.cs_o;                     // LM_UPADD: [sc | sh | sc2 | sh2][cs_i]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n = lane & 15, g = lane >> 4, qq = n >> 2, pp = n & 3;
    const int cs_i = a.cs_i, cs_o = a.cs_o, nci8 = cs_i / 8, nco8 = cs_o / 8;
    for (int i = t; i < 3 * cs_o; i += 256) s_coef[i] = a.coef[i];
    stage_affine_table(a.x, s_aff);

    // dgrad weight fragments (<= 4 input-channel tiles x <= 2 k-steps), in registers for the whole kernel
    f16x8 wf[NF][NSK];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int s = 0; s < NSK; ++s) {
            const bool live = f < a.cit_n && s < a.n_pass;
            wf[f][s] = live ? *reinterpret_cast<const f16x8 *>(a.wpk + ((size_t)(f * a.n_pass + s) * 64 + lane) * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    // weight-gradient accumulators of this wave: input-channel tile `wave` x 4 output tiles, + the bias row (wave 0)
    // (SMALL: input-channel tile wave & 1, k-steps 2 (wave >> 1) .. + 1 of every pixel tile)
    f32x4 accw[NF], accb = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int o = 0; o < NF; ++o) accw[o] = f32x4{0, 0, 0, 0};
    const int wci = SMALL ? (wave & 1) : wave, kk0 = SMALL ? 2 * (wave >> 1) : 0;
    constexpr int NKK = SMALL ? 2 : 4;

    // staging: item i = t + 256 k <-> (pixel i / CH, chunk i % CH) of the tile, the same for dy / z (chunks of cs_o) and x (of cs_i)
    f16x8 r_dy[NS], r_z[NS];
    RawChunk<LM> r_x[NS];
    unsigned ok_px = 0;
    const long long hw = (long long)a.H * a.W;
    auto issue = [&](long long p0) {                        // unconditional, clamped loads
        ok_px = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = t + 256 * k, pl = i >> CSH, c8 = i & (CH - 1);
            const long long px = p0 + pl;
            const bool in = px < a.n_pix;
            const long long pc = in ? px : a.n_pix - 1;
            ok_px |= (in ? 1u : 0u) << k;
            const int co8 = c8 < nco8 ? c8 : 0, ci8 = c8 < nci8 ? c8 : 0;
            r_dy[k] = *reinterpret_cast<const f16x8 *>(a.dy + pc * cs_o + co8 * 8);
            r_z[k] = *reinterpret_cast<const f16x8 *>(a.z + pc * cs_o + co8 * 8);
            if constexpr (LM == LM_RAW) {
                r_x[k].v[0] = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const f16 *>(a.x.in) + pc * cs_i + ci8 * 8);
            } else {
                const int b = (int)(pc / hw);
                const int rem = (int)(pc - (long long)b * hw);
                const int y = rem / a.W, x = rem - y * a.W;
                raw_load<LM>(a.x, b, y, x, a.H, a.W, ci8, r_x[k]);
            }
        }
    };

    const long long n_tiles = (a.n_pix + NPX - 1) / NPX;
    long long tile = blockIdx.x;
    issue((tile < n_tiles ? tile : n_tiles - 1) * NPX);
    __syncthreads();                                        // coefficient / affine tables visible
    while (tile < n_tiles) {
        // ---- registers -> LDS slices: dA = (A dy + B z + C)[z > 0], x ---------------------------------------------------
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = t + 256 * k, pl = i >> CSH, c8 = i & (CH - 1);
            const bool in = (ok_px >> k) & 1u;
            f16x8 d = {0, 0, 0, 0, 0, 0, 0, 0}, xv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (in && c8 < nco8) {
                const float *A = s_coef + c8 * 8, *Bc = A + cs_o, *Cc = Bc + cs_o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float zf = (float)r_z[k][j];
                    d[j] = zf > 0.f ? (f16)(A[j] * (float)r_dy[k][j] + Bc[j] * zf + Cc[j]) : (f16)0.f;
                }
            }
            if (in && c8 < nci8) xv = raw_transform<LM>(r_x[k], s_aff, cs_i, c8, a.x.cin, a.x.u8_div);
            *reinterpret_cast<f16x8 *>(s_d + ((c8 >> 1) * NPX + pl) * H16 + (c8 & 1) * 8) = d;
            *reinterpret_cast<f16x8 *>(s_x + ((c8 >> 1) * NPX + pl) * H16 + (c8 & 1) * 8) = xv;
        }
        __syncthreads();
        const long long next = tile + gridDim.x;
        issue((next < n_tiles ? next : tile) * NPX);        // in flight during the MFMAs (the last one re-reads this tile)

        // ---- dX^T [ci][pixel] = W^T-fragments . dA^T: this wave's 2 pixel groups x all input-channel tiles -----------------
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int pix = (wave * 2 + pg) * 16 + n;
            f32x4 dacc[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) dacc[f] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NSK; ++s) {
                if (s < a.n_pass) {
                    int c8 = s * a.nc8p + g;
                    if (g >= a.nc8p || c8 >= nco8) c8 = 0;                    // zero weights there: any finite chunk
                    const f16x8 bf = *reinterpret_cast<const f16x8 *>(s_d + ((c8 >> 1) * NPX + pix) * H16 + (c8 & 1) * 8);
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        if (f < a.cit_n) dacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f][s], bf, dacc[f], 0, 0, 0);
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (f < a.cit_n) {
                    f16x4 v;
                    const f16x4 xm = *reinterpret_cast<const f16x4 *>(s_x + (f * NPX + pix) * H16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (!MASK || (float)xm[r] > 0.f) ? (f16)dacc[f][r] : (f16)0.f;
                    *reinterpret_cast<f16x4 *>(s_o + pix * 72 + f * 16 + 4 * g) = v;
                }
            }
        }
        // ---- dW [ci][co] += x^T . dA over the tile's 4 k-steps of 32 pixels: this wave's input-channel tile -----------------
        if (wci < a.cit_n) {
#pragma unroll
            for (int kq = 0; kq < NKK; ++kq) {
                const int kk = kk0 + kq;
                const int row = 2 * kk + (g >> 1), xx = 4 * (g & 1) + qq;      // k-slot <-> pixel map of wgrad_mfma_kernel
                const f16 *pa = s_x + (wci * NPX + row * 16 + xx) * H16 + 4 * pp;
                const h4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pa));
                const h4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pa + 8 * H16));
                f16x8 af;
#pragma unroll
                for (int e = 0; e < 4; ++e) { af[e] = (f16)a0[e]; af[4 + e] = (f16)a1[e]; }
#pragma unroll
                for (int o = 0; o < NF; ++o) {
                    if (o < a.cot_n) {
                        const f16 *pb = s_d + (o * NPX + row * 16 + xx) * H16 + 4 * pp;
                        const h4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb));
                        const h4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb + 8 * H16));
                        f16x8 bf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { bf[e] = (f16)b0[e]; bf[4 + e] = (f16)b1[e]; }
                        accw[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, accw[o], 0, 0, 0);
                        if (wci == 0) {                      // column sums of dA -> bias gradient: A = ones in row o
                            f16x8 e1;
#pragma unroll
                            for (int j = 0; j < 8; ++j) e1[j] = (f16)(n == o ? 1.0f : 0.0f);
                            accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(e1, bf, accb, 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();                                    // s_o complete; slice reads done
        // ---- dX tile out: 16-byte chunks, full lines ---------------------------------------------------------------------
        {
            const long long p0 = tile * NPX;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = t + 256 * k, pl = i >> CSH, c8 = i & (CH - 1);
                if (c8 < nci8 && p0 + pl < a.n_pix)
                    *reinterpret_cast<f16x8 *>(a.dx + (p0 + pl) * cs_i + c8 * 8) = *reinterpret_cast<const f16x8 *>(s_o + pl * 72 + c8 * 8);
            }
        }
        __syncthreads();                                    // before the next tile overwrites the slices / s_o
        tile = next;
    }
    // ---- this workgroup's weight-gradient partial rows: [pair = cit * cot_n + cot][tap 0 | bias][256] -----------------------
    if constexpr (SMALL) {      // two partial sums per input-channel tile (waves w and w + 2): combined in that order through LDS
        float *s_acc = reinterpret_cast<float *>(smem);      // [2 tiles][NF + 1][256]; the slices are dead (last barrier of the loop)
        if (wave >= 2) {
#pragma unroll
            for (int o = 0; o < NF; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[((wci * (NF + 1)) + o) * 256 + r * 64 + lane] = accw[o][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_acc[((wci * (NF + 1)) + NF) * 256 + r * 64 + lane] = accb[r];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int o = 0; o < NF; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) accw[o][r] += s_acc[((wci * (NF + 1)) + o) * 256 + r * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) accb[r] += s_acc[((wci * (NF + 1)) + NF) * 256 + r * 64 + lane];
        }
    }
    if (wave < a.cit_n && (!SMALL || wave < 2)) {
        float *wp = a.wg_partial + (size_t)blockIdx.x * a.cit_n * a.cot_n * 2 * 256;
#pragma unroll
        for (int o = 0; o < NF; ++o) {
            if (o < a.cot_n) {
                float *dst = wp + ((size_t)(wave * a.cot_n + o) * 2) * 256 + lane;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * 64] = accw[o][r];
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[256 + r * 64] = (r == 0 && lane < 16) ? accb[o] : 0.f;
                }
            }
        }
    }
    IMK_STAMP_END(1);
}

bool bwd1_env_on() {
    static const bool on = []() { const char *e = getenv("IMK_BWD1X1"); return !(e && e[0] == '0'); }();
    return on;
}

}  // namespace

int imk_bwd1x1_rows(long long n_pix, int cs_in, int cs_out) {
    const long long n_tiles = (n_pix + 127) / 128;
    static const int cap_small = []() { const char *e = getenv("IMK_BWD1X1_ROWS"); return e ? atoi(e) : 256 * BWD1_SMALL_WGS; }();
    const int cap = (cs_in <= 32 && cs_out <= 32) ? cap_small : 512;   // 3 workgroups per compute unit in the <= 32-channel form (4 fit and measured slower: more partial rows), 2 above
    return (int)(n_tiles < cap ? n_tiles : cap);
}

// Conv1x1 with 17-64 (padded) channels on both sides, reading a plain fp16 tensor (the ReLU mask of its dgrad) or upsample + add
bool imk_bwd1x1_ok(int lmode, int cs_in, int cs_out, bool masked) {
    if (!bwd1_env_on()) return false;
    if (cs_in < 24 || cs_in > 64 || cs_out < 24 || cs_out > 64) return false;
    if (lmode == LM_RAW) return masked;
    return lmode == LM_UPADD && !masked;
}

int imk_launch_bwd1x1(const ImkInput &x, const f16 *dy, const f16 *z, const float *coef, const f16 *wpk_bwd, f16 *dx,
                      float *wg_partial, int B, int H, int W, int cout, hipStream_t stream) {
    Bwd1Args a{};
    a.x = x; a.dy = dy; a.z = z; a.coef = coef; a.wpk = wpk_bwd; a.dx = dx; a.wg_partial = wg_partial;
    a.B = B; a.H = H; a.W = W; a.cin = x.cin; a.cs_i = x.cs_in; a.cout = cout; a.cs_o = imk_pad8(cout);
    a.nc8o = a.cs_o / 8; a.nc8p = imk_pass_chunks(a.nc8o); a.n_pass = imk_cdiv_d(a.nc8o, a.nc8p);
    a.cit_n = (a.cs_i + 15) / 16; a.cot_n = (a.cs_o + 15) / 16;
    a.n_pix = (long long)B * H * W;
    if (a.n_pass > 2 || a.cit_n > 4 || a.cot_n > 4) return IMK_EUNSUPPORTED;
    const int grid = imk_bwd1x1_rows(a.n_pix, a.cs_i, a.cs_o);
    const bool small = a.cit_n <= 2 && a.cot_n <= 2 && a.n_pass == 1;
    const size_t lds = (size_t)((small ? 4 : 8) * 128 * WG_STRIDE_H + 128 * 72) * sizeof(f16) + (3 * (size_t)a.cs_o + 4 * (size_t)a.cs_i) * sizeof(float);
    const double px = (double)a.n_pix;
    const double bytes = px * a.cs_o * 4 + px * a.cs_i * 2 * (x.lmode == LM_UPADD ? 1.25 : 1.0) + px * a.cs_i * 2;
    ImkProfScope prof(PF_WGRAD_GEMM, bytes, stream, 4.0 * px * x.cin * cout);
    if (x.lmode == LM_RAW) { if (small) imk_klaunch(bwd1x1_kernel<LM_RAW, true>, dim3(grid), dim3(256), lds, stream, a); else imk_klaunch(bwd1x1_kernel<LM_RAW, false>, dim3(grid), dim3(256), lds, stream, a); }
    else if (x.lmode == LM_UPADD) { if (small) imk_klaunch(bwd1x1_kernel<LM_UPADD, true>, dim3(grid), dim3(256), lds, stream, a); else imk_klaunch(bwd1x1_kernel<LM_UPADD, false>, dim3(grid), dim3(256), lds, stream, a); }
    else return IMK_EUNSUPPORTED;
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}
