/* CPU oracle for the Inconsistency-Mask arithmetic, plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * Scalar restatement of the reference's numpy IM chain, used (a) to cross-check the numpy
 * oracle (oracle/im_oracle.py) and the golden vectors under tests/golden/, and (b) as the
 * "port" CPU baseline of the IM stage in bench.py.  Never linked into libimk.so.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py runs every golden case through this file.
 *
 * Reference lines followed (relative to /root/reference):
 *   functions.py:3104-3120  binary: s = sum of votes; final = (s==N); im = (s!=0 && s!=N)
 *   functions.py:3157       ISIC vote  = p >  thr        functions.py:3187-3189  HeLa vote = p >= thr
 *   functions.py:3195-3200  HeLa: combined IM = max over channels, im_size = sum over channels
 *   functions.py:3123-3137  multiclass: agree = all(label_n == label_0); final = agree ? label_0 : 0
 *   functions.py:3225       label = argmax over classes, lowest index wins ties
 *   functions.py:2867-2874  blocking: image[im>0] = 0, mask[im>0] = 0
 *
 * Layouts: preds [N][H*W][K] float32 (one image), image [H*W][C] u8, outputs [K][H*W] / [H*W] u8.
 */
#include <stdint.h>
#include <stddef.h>

/* Binary / HeLa.  final_out [Kb][HW], im_out [HW] (max over channels), sizes [Kb] each. */
void oracle_im_binary(const float *preds, int n_models, int hw, int kb, float thr, int cmp_ge,
                      const uint8_t *img, int c, int block_in, int block_out,
                      uint8_t *img_out, uint8_t *final_out, uint8_t *im_out,
                      int64_t *im_size, int64_t *pred_size)
{
    for (int k = 0; k < kb; ++k) { im_size[k] = 0; pred_size[k] = 0; }
    for (int p = 0; p < hw; ++p) {
        int any_mixed = 0;
        for (int k = 0; k < kb; ++k) {
            int s = 0;
            for (int m = 0; m < n_models; ++m) {
                float v = preds[((size_t)m * hw + p) * kb + k];
                s += cmp_ge ? (v >= thr) : (v > thr);   /* NaN -> 0 either way */
            }
            int fg = (s == n_models);
            int mixed = (s != 0) && (s != n_models);
            final_out[(size_t)k * hw + p] = fg ? 255 : 0;
            pred_size[k] += fg;
            im_size[k] += mixed;
            any_mixed |= mixed;
        }
        im_out[p] = any_mixed ? 255 : 0;
    }
    for (int p = 0; p < hw; ++p) {
        int hit = im_out[p] > 0;
        if (img && img_out)
            for (int ch = 0; ch < c; ++ch)
                img_out[(size_t)p * c + ch] = (block_in && hit) ? 0 : img[(size_t)p * c + ch];
        if (block_out && hit)
            for (int k = 0; k < kb; ++k) final_out[(size_t)k * hw + p] = 0;
    }
}

/* Multiclass.  presence [N][K] gets 1 for every class a model predicts somewhere. */
void oracle_im_multiclass(const float *probs, int n_models, int hw, int k_classes,
                          const uint8_t *img, int c, int block_in, int block_out,
                          uint8_t *img_out, uint8_t *final_out, uint8_t *im_out,
                          int64_t *im_size, uint8_t *presence)
{
    *im_size = 0;
    for (int i = 0; i < n_models * k_classes; ++i) presence[i] = 0;
    for (int p = 0; p < hw; ++p) {
        int first = 0, agree = 1;
        for (int m = 0; m < n_models; ++m) {
            const float *v = probs + ((size_t)m * hw + p) * k_classes;
            int best = 0;
            for (int k = 1; k < k_classes; ++k)
                if (v[k] > v[best]) best = k;            /* strict > keeps the lowest index on ties */
            presence[m * k_classes + best] = 1;
            if (m == 0) first = best; else agree &= (best == first);
        }
        uint8_t lab = agree ? (uint8_t)first : 0;
        uint8_t im = agree ? 0 : 255;
        *im_size += !agree;
        im_out[p] = im;
        final_out[p] = (block_out && im) ? 0 : lab;
        if (img && img_out)
            for (int ch = 0; ch < c; ++ch)
                img_out[(size_t)p * c + ch] = (block_in && im) ? 0 : img[(size_t)p * c + ch];
    }
}
