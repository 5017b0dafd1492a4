#!/usr/bin/env python3
"""AddressSanitizer + UBSan run of libimk's HOST code (csrc/imk_png.cpp, csrc/imk_geom.cpp: no HIP in them) on the CPU -- GPU sanitizers
are not available on this pool.  Builds the two files with g++ -fsanitize=address,undefined into build/san/, re-runs itself with the
sanitizer runtime preloaded, and drives:
  * the PNG decoder with valid files of every colour type / bit depth it takes and with MUTATED files (bit flips, truncations, spliced
    chunks, forged IHDR sizes, forged chunk lengths): every call must return (0 or an error code) without a sanitizer report, and a
    successful decode must never write past the capacity it was given (guard bytes behind the output buffer);
  * the encoder at the size extremes and with too-small output capacities;
  * the HeLa position geometry (imk_pos_contours / imk_mod_pos_size / imk_cell_count) on random masks, 1-pixel-wide images, full and empty
    images and too-small result capacities.
    python tools/host_sanitize.py [n_mutations]        exit status 0 = clean
Reference paths these replace: cv2.imread / cv2.imwrite (functions.py:2846, 2885-2887), get_pos_contours / mod_pos_size / get_cell_count
(functions.py:6181-6371)."""
import ctypes
import glob
import io
import os
import struct
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "build", "san")
LIB = os.path.join(SAN, "libimk_host_san.so")
SRC = [os.path.join(ROOT, "inconsistencymasks_amd", "csrc", f) for f in ("imk_png.cpp", "imk_geom.cpp")]


def build():
    os.makedirs(SAN, exist_ok=True)
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-fno-omit-frame-pointer", "-fPIC", "-shared", "-o", LIB] + SRC + ["-lz"])


def relaunch():
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(asan), IMK_SAN_CHILD="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    return subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)


def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def make_png(h, w, color_type, depth, rng, interlace=0, palette=True):
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    row_bytes = (w * ch * depth + 7) // 8
    raw = bytearray()
    for r in range(h):
        raw.append(r % 5)
        raw += rng.integers(0, 256, row_bytes).astype("uint8").tobytes()
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, interlace))
    if color_type == 3 and palette:
        out += chunk(b"PLTE", rng.integers(0, 256, 3 * (1 << depth)).astype("uint8").tobytes())
    z = zlib.compress(bytes(raw), 6)
    half = len(z) // 2
    return out + chunk(b"IDAT", z[:half]) + chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")


def main():
    import numpy as np
    n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    lib = ctypes.CDLL(LIB)
    i64, cp = ctypes.c_int64, ctypes.c_void_p
    lib.imk_png_decode.argtypes = [cp, i64, ctypes.c_int, cp, i64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.imk_png_encode.argtypes = [cp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, cp, i64, ctypes.POINTER(i64)]
    lib.imk_pos_contours.argtypes = [cp, ctypes.c_int, ctypes.c_int, ctypes.c_int, cp, ctypes.c_int]
    lib.imk_mod_pos_size.argtypes = [cp] + [ctypes.c_int] * 6 + [cp]
    lib.imk_cell_count.argtypes = [cp, ctypes.c_int, cp, cp, ctypes.c_int, ctypes.c_int, ctypes.c_int, cp]
    rng = np.random.default_rng(11)
    GUARD = 64
    stats = {"decoded": 0, "declined": 0}

    def decode(data, want_c, cap_pixels):
        """cap_pixels = capacity handed to the decoder; the buffer has GUARD bytes of 0xA5 behind it that must survive"""
        buf = np.full(cap_pixels + GUARD, 0xA5, np.uint8)
        src = np.frombuffer(bytes(data), np.uint8).copy()       # an exact-size heap copy: ASan sees reads past the end
        h, w = ctypes.c_int(0), ctypes.c_int(0)
        rc = lib.imk_png_decode(src.ctypes.data, src.size, want_c, buf.ctypes.data, cap_pixels, ctypes.byref(h), ctypes.byref(w))
        assert (buf[cap_pixels:] == 0xA5).all(), "decoder wrote past its capacity"
        if rc == 0:
            assert 0 < h.value and 0 < w.value and h.value * w.value * want_c <= cap_pixels, (h.value, w.value, cap_pixels)
            stats["decoded"] += 1
        else:
            stats["declined"] += 1
        return rc, h.value, w.value, buf

    # 1. valid files of every colour type / depth (accepted or declined, never a report), exact and too-small capacities
    seeds = []
    for ct, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for d in depths:
            for (h, w) in ((1, 1), (3, 5), (17, 31), (64, 64)):
                for il in (0, 1):
                    p = make_png(h, w, ct, d, rng, interlace=il)
                    seeds.append(p)
                    for want in (1, 3):
                        decode(p, want, h * w * want)
                        decode(p, want, max(0, h * w * want - 1))
                        decode(p, want, 0)
    seeds.append(make_png(8, 8, 3, 8, rng, palette=False))          # palette image without PLTE
    try:
        from PIL import Image
        for mode, shape in (("L", (40, 50)), ("RGB", (40, 50, 3)), ("RGBA", (12, 9, 4)), ("P", (20, 20))):
            bio = io.BytesIO()
            Image.fromarray(rng.integers(0, 256, shape).astype("uint8"), mode=mode).save(bio, format="PNG", optimize=True)
            seeds.append(bio.getvalue())
    except ImportError:
        pass
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.png")))[:8]:
        seeds.append(open(f, "rb").read())

    # 2. mutations
    for it in range(n_mut):
        s = bytearray(seeds[it % len(seeds)])
        kind = int(rng.integers(0, 8))
        if kind == 0:                                   # bit flips anywhere
            for _ in range(int(rng.integers(1, 8))):
                s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                 # truncation
            s = s[:int(rng.integers(0, len(s)))]
        elif kind == 2:                                 # forged IHDR size (CRC fixed up so the parser goes on)
            w, h = [int(x) for x in rng.choice([0, 1, 2, 7, 255, 256, 65535, 65536, 1 << 20, (1 << 31) - 1, (1 << 32) - 1], 2)]
            ihdr = struct.pack(">II", w, h) + bytes(s[24:29])
            s[8:33] = chunk(b"IHDR", ihdr)
        elif kind == 3:                                 # forged chunk length somewhere
            pos = 8
            offs = []
            while pos + 8 <= len(s):
                offs.append(pos)
                pos += 12 + struct.unpack(">I", s[pos:pos + 4])[0]
            o = offs[int(rng.integers(0, len(offs)))]
            s[o:o + 4] = struct.pack(">I", int(rng.choice([0, 1, len(s), (1 << 31) - 1, (1 << 32) - 1, (1 << 31)])))
        elif kind == 4:                                 # bytes of the compressed stream replaced (CRC left wrong or fixed)
            i = s.find(b"IDAT")
            if i > 0:
                n = struct.unpack(">I", s[i - 4:i])[0]
                for _ in range(int(rng.integers(1, 6))):
                    if n > 0:
                        s[i + 4 + int(rng.integers(0, n))] = int(rng.integers(0, 256))
                if rng.integers(0, 2):
                    s[i + 4 + n:i + 8 + n] = struct.pack(">I", zlib.crc32(bytes(s[i:i + 4 + n])) & 0xFFFFFFFF)
        elif kind == 5:                                 # a valid zlib stream of the wrong length behind a valid header
            i = s.find(b"IDAT")
            if i > 0:
                z = zlib.compress(rng.integers(0, 256, int(rng.integers(0, 300))).astype("uint8").tobytes())
                s = s[:i - 4] + chunk(b"IDAT", z) + chunk(b"IEND", b"")
        elif kind == 6:                                 # the UNCOMPRESSED rows mutated (filter bytes included), re-deflated: a valid stream of the right length
            pos, z, head, ok = 8, b"", None, True
            while pos + 12 <= len(s):
                n = struct.unpack(">I", s[pos:pos + 4])[0]
                if s[pos + 4:pos + 8] == b"IDAT":
                    z += bytes(s[pos + 8:pos + 8 + n])
                    head = pos if head is None else head
                pos += 12 + n
            try:
                raw = bytearray(zlib.decompress(z))
            except zlib.error:
                ok = False
            if ok and raw and head is not None:
                for _ in range(int(rng.integers(1, 10))):
                    raw[int(rng.integers(0, len(raw)))] = int(rng.integers(0, 256))
                s = s[:head] + chunk(b"IDAT", zlib.compress(bytes(raw), 1)) + chunk(b"IEND", b"")
        else:                                           # random garbage with a PNG signature
            s = bytearray(b"\x89PNG\r\n\x1a\n") + rng.integers(0, 256, int(rng.integers(0, 200))).astype("uint8").tobytes()
        if kind in (0, 3) and rng.integers(0, 2):       # every chunk's CRC made right again: the mutation reaches the parser behind the check
            pos = 8
            while pos + 12 <= len(s):
                n = struct.unpack(">I", s[pos:pos + 4])[0]
                if pos + 12 + n > len(s):
                    break
                s[pos + 8 + n:pos + 12 + n] = struct.pack(">I", zlib.crc32(bytes(s[pos + 4:pos + 8 + n])) & 0xFFFFFFFF)
                pos += 12 + n
        want = 1 if it & 1 else 3
        decode(s, want, int(rng.choice([0, 1, 64, 4096, 64 * 64 * 3])))

    # 3. encoder: round trips, size extremes, too-small capacities
    for (h, w, c) in ((1, 1, 1), (1, 1, 3), (1, 4097, 3), (2049, 1, 1), (37, 53, 3), (256, 256, 1)):
        px = rng.integers(0, 256, (h, w, c)).astype("uint8")
        for level in (0, 1, 6, 9):
            for cap in (h * w * c * 2 + 4096, 64, 8, 0):
                out = np.full(cap + GUARD, 0xA5, np.uint8)
                n = i64(0)
                rc = lib.imk_png_encode(px.ctypes.data, h, w, c, level, out.ctypes.data, cap, ctypes.byref(n))
                assert (out[cap:] == 0xA5).all(), "encoder wrote past its capacity"
                if rc == 0:
                    assert 0 < n.value <= cap
                    rc2, hh, ww, buf = decode(out[:n.value].tobytes(), c, h * w * c)
                    assert rc2 == 0 and (hh, ww) == (h, w) and np.array_equal(buf[:h * w * c].reshape(h, w, c), px)
                else:
                    assert cap < h * w * c * 2 + 4096
    for bad in ((0, 5, 3), (5, 0, 1), (-1, 5, 1), (4, 4, 2), (4, 4, 5)):
        out = np.zeros(1024, np.uint8)
        n = i64(0)
        assert lib.imk_png_encode(out.ctypes.data, bad[0], bad[1], bad[2], 6, out.ctypes.data, 1024, ctypes.byref(n)) != 0

    # 4. geometry
    n_geo = 0
    for (h, w) in ((1, 1), (1, 64), (64, 1), (2, 2), (3, 3), (17, 31), (64, 64), (256, 256)):
        for fill in ("empty", "full", "noise", "blobs", "border"):
            img = np.zeros((h, w), np.uint8)
            if fill == "full":
                img[:] = 255
            elif fill == "noise":
                img = (rng.random((h, w)) > 0.5).astype(np.uint8) * 255
            elif fill == "blobs":
                yy, xx = np.mgrid[0:h, 0:w]
                for _ in range(12):
                    cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(1, 9)
                    img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 255
                img[(rng.random((h, w)) > 0.97)] = 0           # holes
            elif fill == "border":
                img[0, :] = img[-1, :] = 255
                img[:, 0] = img[:, -1] = 255
            img = np.ascontiguousarray(img)
            for ek in (0, 1, 3, 5):
                for cap in (0, 1, 4, 4096):
                    xy = np.full(2 * cap + GUARD, -7, np.int32)
                    n = lib.imk_pos_contours(img.ctypes.data, h, w, ek, xy.ctypes.data, cap)
                    assert (xy[2 * cap:] == -7).all(), "imk_pos_contours wrote past its capacity"
                    n_geo += 1
                    if n > 0 and cap >= n:
                        pts = xy[:2 * n].reshape(n, 2)
                        assert (pts[:, 0] >= 1).all() and (pts[:, 0] <= w).all() and (pts[:, 1] >= 1).all() and (pts[:, 1] <= h).all()   # centroid + 1, as the reference
                        alive = np.ascontiguousarray((rng.random((h, w)) > 0.5).astype(np.uint8) * 255)
                        dead = np.ascontiguousarray((rng.random((h, w)) > 0.5).astype(np.uint8) * 255)
                        counts = np.full(3 + 8, -7, np.int32)
                        for mr in (0, 1, 3, 40):
                            counts[:] = -7
                            rc = lib.imk_cell_count(pts.ctypes.data, n, alive.ctypes.data, dead.ctypes.data, h, w, mr, counts.ctypes.data)
                            assert (counts[3:] == -7).all()
                            assert (rc == 0 and counts[:3].sum() == n) or (rc != 0 and (mr <= 0 or h < 2 * mr or w < 2 * mr)), (rc, counts, n, mr)
            for (mx, mn, lone, blur) in ((9, 3, 30, 1), (1, 1, 0, 0), (40, 2, 5, 1)):
                out = np.full(h * w + GUARD, 0xA5, np.uint8)
                lib.imk_mod_pos_size(img.ctypes.data, h, w, mx, mn, lone, blur, out.ctypes.data)
                assert (out[h * w:] == 0xA5).all(), "imk_mod_pos_size wrote past the image"
                n_geo += 1
    print(f"host sanitizer run clean: {len(seeds)} seed files, {n_mut} mutations ({stats['decoded']} decodes accepted, "
          f"{stats['declined']} declined), encoder round trips, {n_geo} geometry calls")


if __name__ == "__main__":
    if os.environ.get("IMK_SAN_CHILD") != "1":
        build()
        sys.exit(relaunch())
    main()
