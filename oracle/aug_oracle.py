"""CPU restatement (numpy) of the reference's Noisy-Student augmentation chain -- TEST INFRASTRUCTURE ONLY.

Follows augment_image_and_mask (functions.py:2779-2826), add_noise_and_blur (:1481-1506) and add_noise (:1463-1478) for
one image with the random draws made explicit (the reference draws them from Python's / numpy's unseeded global
streams).  The noise arithmetic (add, clip, dtype) is PINNED to outputs of the reference's add_noise
(tests/golden/augment.npz).  PARITY UNPINNED for the OpenCV pieces: cv2 is not installed here, so GaussianBlur((k,k),0) is restated from
OpenCV's documented behaviour for 8-bit images (fixed small kernels for sigma = 0, BORDER_REFLECT_101, exact
fixed-point accumulation rounded half up) and convertScaleAbs as saturate(round-half-even(|a*x+b|)).  The flips and
rotations are checked against numpy's flip / rot90 in tests/test_oracle_golden.py.  The noise generator is this
framework's own counter-based hash (see csrc/imk_aug.hip); it has no counterpart in the reference to pin to.
"""
import numpy as np

_GAUSS = {3: np.array([16, 32, 16], np.int64), 5: np.array([4, 16, 24, 16, 4], np.int64),
          7: np.array([2, 7, 14, 18, 14, 7, 2], np.int64)}    # x/64: OpenCV's small_gaussian_tab


def hash32(x):
    x = np.asarray(x, np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def geometric(a, flip_v, flip_h, rot):
    """cv2.flip(a,0) / cv2.flip(a,1) / cv2.rotate  (functions.py:2795-2818)."""
    if flip_v:
        a = a[::-1]
    if flip_h:
        a = a[:, ::-1]
    if rot == 1:
        a = np.rot90(a, k=-1)        # ROTATE_90_CLOCKWISE
    elif rot == 2:
        a = np.rot90(a, k=2)
    elif rot == 3:
        a = np.rot90(a, k=1)         # ROTATE_90_COUNTERCLOCKWISE
    return np.ascontiguousarray(a)


def convert_scale_abs(a, alpha, beta):
    """cv2.convertScaleAbs (functions.py:2824): float32 arithmetic, round half to even, saturate."""
    v = np.abs(a.astype(np.float32) * np.float32(alpha) + np.float32(beta))
    return np.minimum(np.rint(v), 255).astype(np.uint8)


def gaussian_blur(a, k):
    """cv2.GaussianBlur(a, (k,k), 0) for uint8 (functions.py:1495-1501)."""
    if k <= 1:
        return a
    w = _GAUSS[k]
    r = k // 2
    p = np.pad(a.astype(np.int64), ((r, r), (r, r), (0, 0)), mode="reflect")     # BORDER_REFLECT_101
    H, W = a.shape[:2]
    rows = sum(w[i] * p[:, i:i + W] for i in range(k))
    acc = sum(w[i] * rows[i:i + H] for i in range(k))
    return ((acc + 2048) >> 12).astype(np.uint8)


def noise_field(shape, noise_max, seed):
    """uniform integers in [-m, m) per element (functions.py:1475), from hash(seed, element index)."""
    e = np.arange(int(np.prod(shape)), dtype=np.uint64)
    h = hash32(hash32(np.uint64(seed) ^ np.uint64(0x9E3779B9)) + e)
    return ((h * np.uint64(2 * noise_max)) >> np.uint64(32)).astype(np.int64).reshape(shape) - noise_max


def apply_noise(image, noise):
    """image + noise, clipped to 0..255 (functions.py:1476-1478) -- pinned by tests/golden/augment.npz, which holds
    outputs of the reference's own add_noise together with the noise field it drew."""
    return np.clip(image.astype(np.int64) + noise, 0, 255).astype(np.uint8)


def augment(image, mask, flip_v, flip_h, rot, bright_on, alpha, beta, blur_k, noise_max, seed):
    """image [H,W,C] u8, mask [H,W,Cm] u8 or None -> (aug_image, aug_mask)."""
    img = geometric(image, flip_v, flip_h, rot)
    msk = geometric(mask, flip_v, flip_h, rot) if mask is not None else None
    if bright_on:
        img = convert_scale_abs(img, alpha, beta)
    img = gaussian_blur(img, blur_k)
    if noise_max > 0:
        img = apply_noise(img, noise_field(img.shape, noise_max, seed))
    return img, msk
