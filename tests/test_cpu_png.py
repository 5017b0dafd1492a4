"""libimk's host PNG codec (csrc/imk_png.cpp; include/imk.h imk_png_*) against Pillow -- an implementation nobody here wrote -- in both
directions: what the native decoder accepts it must decode exactly like Pillow's convert("RGB") / convert("L") (what rounds 1-4
returned), what it declines must reach Pillow through functions.read_png, and every file the native encoder writes must be a PNG
Pillow reads back pixel for pixel.  Host code only: no GPU call is made (the reference uses cv2.imread /
cv2.imwrite, functions.py:2846, 2885-2887)."""
import ctypes
import io
import os
import threading

import numpy as np
import pytest
from PIL import Image

from inconsistencymasks_amd import functions as F
from inconsistencymasks_amd._lib import lib


def _pil_file(tmp_path, name, mode, arr, **save):
    im = Image.fromarray(arr, mode=mode)
    if mode == "P":
        im.putpalette(np.random.default_rng(7).integers(0, 256, 768).astype(np.uint8).tolist())
    p = str(tmp_path / name)
    im.save(p, **save)
    return p


@pytest.mark.parametrize("mode,shape", [("L", (37, 53)), ("RGB", (37, 53, 3)), ("RGBA", (20, 31, 4)), ("LA", (20, 31, 2)), ("P", (25, 40)),
                                        ("RGB", (1, 1, 3)), ("L", (256, 256)), ("RGB", (208, 416, 3))])
@pytest.mark.parametrize("level", [0, 1, 9])
def test_decoder_equals_pillow(tmp_path, mode, shape, level):
    rng = np.random.default_rng(len(mode) * 100 + shape[0])
    arr = rng.integers(0, 256, shape).astype(np.uint8)
    if shape[0] > 100:      # a smooth image: Pillow's encoder then picks Sub / Up / Average / Paeth filters, not only None
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        base = (128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 13.0)).astype(np.int32)
        arr = (base[..., None] + rng.integers(-3, 4, shape if arr.ndim == 3 else shape + (1,))).clip(0, 255).astype(np.uint8).reshape(shape)
    p = _pil_file(tmp_path, f"{mode}.png", mode, arr, compress_level=level)
    for ch in (1, 3):
        got = F.read_png(p, ch)
        assert got.shape == shape[:2] + (ch,) and got.dtype == np.uint8
        assert np.array_equal(got, F._read_png_pillow(p, ch)), (mode, ch)
        # the native path was the one that ran
        h, w = ctypes.c_int(), ctypes.c_int()
        out = np.empty(shape[:2] + (ch,), np.uint8)
        assert lib.imk_png_read_file(os.fsencode(p), ch, out.ctypes.data, out.nbytes, ctypes.byref(h), ctypes.byref(w)) == 0
        assert (h.value, w.value) == shape[:2] and np.array_equal(out, got)


def test_all_five_filter_types_are_decoded(tmp_path):
    """a file whose rows use filter 0, 1, 2, 3, 4 in turn (written by hand with zlib: Pillow chooses filters itself)"""
    import struct
    import zlib
    rng = np.random.default_rng(3)
    h, w, c = 10, 17, 3
    img = rng.integers(0, 256, (h, w, c)).astype(np.uint8)
    raw = bytearray()
    flat = img.reshape(h, w * c).astype(np.int32)
    for r in range(h):
        ft = r % 5
        x, up = flat[r], flat[r - 1] if r else np.zeros(w * c, np.int32)
        left = np.concatenate([np.zeros(c, np.int32), x[:-c]])
        ul = np.concatenate([np.zeros(c, np.int32), up[:-c]])
        if ft == 0: f = x
        elif ft == 1: f = x - left
        elif ft == 2: f = x - up
        elif ft == 3: f = x - ((left + up) >> 1)
        else:
            pa, pb, pc = np.abs(up - ul), np.abs(left - ul), np.abs(left + up - 2 * ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
            f = x - pred
        raw += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    z = zlib.compress(bytes(raw), 6)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", z[:40]) + chunk(b"IDAT", z[40:]) + chunk(b"IEND", b"")
    p = tmp_path / "filters.png"
    p.write_bytes(png)
    assert np.array_equal(np.asarray(Image.open(p)), img)                  # the hand-made file is a valid PNG (two IDAT chunks)
    assert np.array_equal(F.read_png(str(p), 3), img)
    out = np.empty((h, w, 3), np.uint8)
    assert lib.imk_png_decode(png, len(png), 3, out.ctypes.data, out.nbytes, None, None) == 0 and np.array_equal(out, img)


def test_declined_formats_fall_back_to_pillow(tmp_path):
    rng = np.random.default_rng(5)
    a16 = rng.integers(0, 65536, (12, 9)).astype(np.uint16)
    p16 = str(tmp_path / "i16.png")
    Image.fromarray(a16).save(p16)
    bit = rng.integers(0, 2, (12, 9)).astype(bool)
    p1 = str(tmp_path / "bit.png")
    Image.fromarray(bit).save(p1)
    out = np.empty((12, 9, 1), np.uint8)
    for p in (p16, p1):
        assert lib.imk_png_read_file(os.fsencode(p), 1, out.ctypes.data, out.nbytes, None, None) == -2        # IMK_EUNSUPPORTED
        assert np.array_equal(F.read_png(p, 1), F._read_png_pillow(p, 1))
        assert np.array_equal(F.read_png(p, 3), F._read_png_pillow(p, 3))
    # not a PNG at all / truncated: an error from the native call, then Pillow's own exception through read_png
    junk = tmp_path / "junk.png"
    junk.write_bytes(b"not a png")
    assert lib.imk_png_read_file(os.fsencode(str(junk)), 1, out.ctypes.data, out.nbytes, None, None) == -1
    with pytest.raises(Exception):
        F.read_png(str(junk), 1)
    good = tmp_path / "good.png"
    F.write_png(str(good), rng.integers(0, 256, (12, 9, 3)).astype(np.uint8))
    cut = tmp_path / "cut.png"
    cut.write_bytes(good.read_bytes()[:-30])
    assert lib.imk_png_read_file(os.fsencode(str(cut)), 3, np.empty((12, 9, 3), np.uint8).ctypes.data, 12 * 9 * 3, None, None) != 0
    # too small a buffer is refused, not overrun
    assert lib.imk_png_read_file(os.fsencode(str(good)), 3, out.ctypes.data, out.nbytes, None, None) == -3


@pytest.mark.parametrize("shape", [(64, 48), (64, 48, 3), (256, 256, 3), (1, 1), (3, 5, 1), (208, 416), (7, 1, 3)])
def test_encoder_output_is_read_back_by_pillow(tmp_path, shape):
    rng = np.random.default_rng(sum(shape))
    for kind in ("noise", "smooth", "mask"):
        if kind == "noise":
            a = rng.integers(0, 256, shape).astype(np.uint8)
        elif kind == "smooth":
            yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
            a = np.broadcast_to((128 + 90 * np.sin(xx / 7.0 + yy / 11.0))[(...,) + (None,) * (len(shape) - 2)], shape).astype(np.uint8).copy()
        else:
            a = ((rng.random(shape[:2]) > 0.5).astype(np.uint8) * 255).reshape(shape[:2] + (1,) * (len(shape) - 2))
            a = np.broadcast_to(a, shape).copy()
        p = str(tmp_path / f"{kind}.png")
        F.write_png(p, a)
        back = np.asarray(Image.open(p))
        want = a[..., 0] if a.ndim == 3 and a.shape[2] == 1 else a
        assert back.dtype == np.uint8 and np.array_equal(back, want), (shape, kind)
        assert np.array_equal(F.read_png(p, 3 if (a.ndim == 3 and a.shape[2] == 3) else 1).reshape(a.shape), a)
        with Image.open(p) as im:
            im.verify()                        # CRCs and structure


def test_codec_scales_without_the_interpreter_lock(tmp_path):
    """many threads through the package's own pool: all files complete and correct (the lock-free property itself is a rate,
    reported by bench.py's png_io; here: thread safety)"""
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, (64, 64, 3)).astype(np.uint8) for _ in range(96)]
    for i, a in enumerate(imgs):
        F.write_png_async(str(tmp_path / f"{i:03d}.png"), a)
    F.flush_writes()
    res = [None] * len(imgs)
    def work(i):
        res[i] = F.read_png(str(tmp_path / f"{i:03d}.png"), 3)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(imgs))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert all(np.array_equal(a, b) for a, b in zip(imgs, res))


def test_pillow_can_be_forced(tmp_path, monkeypatch):
    monkeypatch.setattr(F, "_NATIVE_PNG", False)
    a = np.random.default_rng(1).integers(0, 256, (20, 30, 3)).astype(np.uint8)
    p = str(tmp_path / "p.png")
    F.write_png(p, a)
    assert np.array_equal(F.read_png(p, 3), a)


def test_stack_reader_equals_file_by_file_reads(tmp_path):
    """read_png_stack (every pool thread decodes straight into its rows of one array) against read_png per file: more files than the
    pool's one-future-per-item limit (so the sliced path runs), a 16-bit file among them (Pillow's route into the row), RGB and grey,
    and a file of another size is an error, not a silently reshaped row"""
    rng = np.random.default_rng(12)
    n = 4 * F._IO_THREADS + 37
    paths = []
    for i in range(n):
        p = str(tmp_path / f"{i:04d}.png")
        if i == 5:
            Image.fromarray(rng.integers(0, 65536, (24, 40)).astype(np.uint16)).save(p)      # 16-bit greyscale: declined by libimk
        else:
            F.write_png(p, rng.integers(0, 256, (24, 40, 3)).astype(np.uint8))
        paths.append(p)
    with F._pool() as pool:
        for c in (3, 1):
            got = F.read_png_stack(pool, paths, c)
            assert got.shape == (n, 24, 40, c) and got.dtype == np.uint8 and got.flags["C_CONTIGUOUS"]
            for i in (0, 1, 5, 6, n // 2, n - 1):
                assert np.array_equal(got[i], F.read_png(paths[i], c)), (c, i)
        assert np.array_equal(F.read_png_stack(pool, paths[:1], 3)[0], F.read_png(paths[0], 3))
        odd = str(tmp_path / "odd.png")
        F.write_png(odd, np.zeros((24, 41, 3), np.uint8))
        with pytest.raises(Exception):
            F.read_png_stack(pool, paths[:3] + [odd], 3)
        # the pool keeps the order of a long list and passes a task's exception on
        assert pool.map(lambda v: v * v, range(1000)) == [v * v for v in range(1000)]
        with pytest.raises(ZeroDivisionError):
            pool.map(lambda v: 1 // (v - 700), range(1000))


def test_forged_header_is_an_error_code_not_a_crash():
    """A PNG whose IHDR claims 2^31 - 1 pixels a side (round 5, found by tools/host_sanitize.py: h * w * 3 wrapped past the capacity check and
    the row buffer's allocation then threw through the C ABI): every entry point returns an error code, for any capacity."""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    for (w, h) in ((2**31 - 1, 2**31 - 1), (2**31 - 1, 1), (1, 2**31 - 1), (65536, 65536), (1 << 20, 1 << 20)):
        png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
               + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""))
        src = np.frombuffer(png, np.uint8).copy()
        for cap in (0, 16, 1 << 16):
            out = np.zeros(max(cap, 1), np.uint8)
            hh, ww = ctypes.c_int(), ctypes.c_int()
            for want in (1, 3):
                rc = lib.imk_png_decode(src.ctypes.data, ctypes.c_int64(src.size), want, out.ctypes.data, ctypes.c_int64(cap),
                                        ctypes.byref(hh), ctypes.byref(ww))
                assert rc != 0
    # the writer validates before it sizes its buffer
    px = np.zeros((4, 4, 3), np.uint8)
    for (h, w, c) in ((0, 4, 3), (4, -1, 3), (4, 4, 2)):
        assert lib.imk_png_write_file(os.fsencode("/tmp/_imk_never_written.png"), px.ctypes.data, h, w, c, 1) != 0
    assert not os.path.exists("/tmp/_imk_never_written.png")


def test_damaged_header_or_palette_chunk_is_refused(tmp_path):
    """CRC-32 of IHDR and PLTE is verified (IDAT is covered by zlib's adler32) -- a bit flip in the size or in the palette
    is an error, as it is in Pillow, not a silently different image; the file API then falls back to Pillow, which raises."""
    rng = np.random.default_rng(3)
    idx = rng.integers(0, 200, (12, 20)).astype(np.uint8)              # > 16 colours: Pillow keeps 8 bits per index
    im = Image.fromarray(idx, mode="P")
    im.putpalette([v for k in range(200) for v in (k, 255 - k, (7 * k) % 256)])
    buf = io.BytesIO()
    im.save(buf, format="PNG")
    good = np.frombuffer(buf.getvalue(), np.uint8).copy()
    out = np.zeros((12, 20, 3), np.uint8)
    dec = lambda a: lib.imk_png_decode(a.ctypes.data, ctypes.c_int64(a.size), 3, out.ctypes.data, ctypes.c_int64(out.nbytes), None, None)
    assert dec(good) == 0 and np.array_equal(out, np.asarray(im.convert("RGB")))
    raw = good.tobytes()
    for where in (raw.index(b"IHDR") + 4 + 3, raw.index(b"PLTE") + 4 + 5):      # low byte of the width; one palette entry
        bad = good.copy()
        bad[where] ^= 0x01
        assert dec(bad) != 0
        p = tmp_path / f"bad_{where}.png"
        p.write_bytes(bad.tobytes())
        with pytest.raises(Exception):
            F.read_png(str(p), 3)


def test_file_writes_are_renamed_into_place(tmp_path):
    """a finished file appears under its final name only; nothing else is left in the directory, and a write into a missing directory
    fails without leaving a temporary file"""
    a = np.random.default_rng(4).integers(0, 256, (16, 24, 3)).astype(np.uint8)
    F.write_png(str(tmp_path / "x.png"), a)
    assert os.listdir(tmp_path) == ["x.png"] and np.array_equal(F.read_png(str(tmp_path / "x.png"), 3), a)
    rc = lib.imk_png_write_file(os.fsencode(str(tmp_path / "missing" / "y.png")), a.ctypes.data, 16, 24, 3, 1)
    assert rc != 0 and os.listdir(tmp_path) == ["x.png"]
