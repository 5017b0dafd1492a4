"""Host wrappers of the fused IM kernels (imk_im_binary / imk_im_multiclass / imk_morph / imk_block_apply).
torch is used for device memory and streams only."""
import torch

from ._lib import check, lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev_u8(x, device):
    if x is None:
        return None
    t = torch.as_tensor(x)
    if t.dtype != torch.uint8:
        raise TypeError("images must be uint8")
    return t.to(device).contiguous()


def im_binary(preds, thr=0.5, cmp_ge=False, img=None, block_in=True, block_out=True):
    """preds float32 [N,B,H,W,Kb] (device) -> dict of device tensors:
    masks [B,Kb,H,W] u8, im [B,H,W] u8, im_size [B,Kb] i64, pred_size [B,Kb] i64, img_out [B,H,W,C] u8|None.
    Semantics: functions.py:3104-3120, 3140-3202, 2867-2874 (see include/imk.h)."""
    if preds.dtype != torch.float32 or preds.dim() != 5 or not preds.is_cuda:
        raise TypeError("preds must be a float32 CUDA tensor [N,B,H,W,Kb]")
    preds = preds.contiguous()
    n, b, h, w, kb = preds.shape
    dev = preds.device
    img = _dev_u8(img, dev)
    c = 0 if img is None else img.shape[-1]
    masks = torch.empty((b, kb, h, w), dtype=torch.uint8, device=dev)
    im = torch.empty((b, h, w), dtype=torch.uint8, device=dev)
    im_size = torch.empty((b, kb), dtype=torch.int64, device=dev)
    pred_size = torch.empty((b, kb), dtype=torch.int64, device=dev)
    img_out = None if img is None else torch.empty_like(img)
    check(lib.imk_im_binary(preds.data_ptr(), n, b, h, w, kb, float(thr), int(bool(cmp_ge)),
                            _ptr(img), c, int(bool(block_in)), int(bool(block_out)),
                            _ptr(img_out), masks.data_ptr(), im.data_ptr(),
                            im_size.data_ptr(), pred_size.data_ptr(), _stream()), "imk_im_binary")
    return {"masks": masks, "im": im, "im_size": im_size, "pred_size": pred_size, "img_out": img_out}


def im_multiclass(probs, img=None, block_in=True, block_out=True, want_presence=True):
    """probs float32 [N,B,H,W,K] (device) -> final [B,H,W] u8, im [B,H,W] u8, im_size [B] i64,
    presence [N,B,K] u8, img_out.  Semantics: functions.py:3123-3137, 3206-3238."""
    if probs.dtype != torch.float32 or probs.dim() != 5 or not probs.is_cuda:
        raise TypeError("probs must be a float32 CUDA tensor [N,B,H,W,K]")
    probs = probs.contiguous()
    n, b, h, w, k = probs.shape
    dev = probs.device
    img = _dev_u8(img, dev)
    c = 0 if img is None else img.shape[-1]
    final = torch.empty((b, h, w), dtype=torch.uint8, device=dev)
    im = torch.empty((b, h, w), dtype=torch.uint8, device=dev)
    im_size = torch.empty((b,), dtype=torch.int64, device=dev)
    presence = torch.empty((n, b, k), dtype=torch.uint8, device=dev) if want_presence else None
    img_out = None if img is None else torch.empty_like(img)
    check(lib.imk_im_multiclass(probs.data_ptr(), n, b, h, w, k, _ptr(img), c,
                                int(bool(block_in)), int(bool(block_out)), _ptr(img_out),
                                final.data_ptr(), im.data_ptr(), im_size.data_ptr(), _ptr(presence),
                                _stream()), "imk_im_multiclass")
    return {"final": final, "im": im, "im_size": im_size, "presence": presence, "img_out": img_out}


def morph(mask, ksize, op):
    """mask u8 [B,H,W] device; op 'erode' | 'dilate' with a ksize x ksize ones kernel (functions.py:2858-2864)."""
    mask = mask.contiguous()
    b, h, w = mask.shape
    out = torch.empty_like(mask)
    check(lib.imk_morph(mask.data_ptr(), out.data_ptr(), b, h, w, int(ksize), 0 if op == "erode" else 1,
                        _stream()), "imk_morph")
    return out


def block_apply(im, img=None, masks=None):
    """In place: img[im>0]=0 (img [B,H,W,C]) and masks[:, m][im>0]=0 (masks [B,M,H,W])."""
    b, h, w = im.shape
    c = 0 if img is None else img.shape[-1]
    m = 0 if masks is None else masks.shape[1]
    check(lib.imk_block_apply(im.data_ptr(), _ptr(img), c, _ptr(masks), m, b, h, w, _stream()), "imk_block_apply")
