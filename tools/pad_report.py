#!/usr/bin/env python3
"""Padded against logical work of every conv of the U-Net at a given width ("alpha = 1.25 is anomalous").

For each conv layer: logical (cin, cout); what the kernels multiply: K padded to (passes x chunks per pass x 8) input channels
(imk_stage.h: imk_pass_chunks, IMK_PASS_CAP = 4) and M padded to the workgroup's output-channel tile (conv_pipe / conv_wide: 16 / 32;
conv_gemm: 64 or 128, imk_gemm.hip: plan_conv_gemm); the weight gradient's 64 x 64 (ci, co) blocks (imk_wgemm.hip).  The last
columns are logical MACs / padded MACs for forward + dgrad and for the weight gradient, and the FLOP-weighted totals per width.

    python tools/pad_report.py [alpha ...]          (default: 1 1.25 1.5 1.75 2; Cityscapes 208 x 416, 35 classes)
"""
import math
import sys

PASS_CAP = 4


def pad8(c):
    return (c + 7) // 8 * 8


def pass_chunks(nc8):
    if nc8 <= PASS_CAP:
        return nc8
    n_pass = -(-nc8 // PASS_CAP)
    return -(-nc8 // n_pass)


def layers(alpha, c_in=3, k=35):
    f = lambda v: int(v * alpha)
    c16, c32, c64, c128, c256 = f(16), f(32), f(64), f(128), f(256)
    t = [("in.c", 1, c_in, c16, 0)]
    for i, (ci, co) in enumerate([(c16, c16), (c16, c32), (c32, c64), (c64, c128)], start=1):
        t += [(f"e{i}.c3", 3, ci, co, i - 1), (f"e{i}.c1", 1, co, co, i - 1)]
    t += [("b.c3", 3, c128, c256, 4), ("b.c1", 1, c256, c128, 4)]
    for j, ci, f1, f2, lv in [(6, c128, c128, c64, 3), (7, c64, c64, c32, 2), (8, c32, c32, c16, 1), (9, c16, c16, c16, 0)]:
        t += [(f"d{j}.ca", 1, ci, f1, lv), (f"d{j}.c3", 3, f1, f1, lv), (f"d{j}.c1", 1, f1, f2, lv)]
    return t


def conv_tile(cin, cout):
    """(kernel family, padded K channels, padded M channels) of one forward / dgrad launch with these logical widths"""
    ci8, co8 = pad8(cin), pad8(cout)
    if ci8 <= 16 and cout <= 16:
        return "conv_pipe", ci8, 8 if cout <= 8 else 16
    if ci8 <= 32 and co8 <= 32:
        return "conv_wide", ci8, 16 * (-(-cout // 16))
    nc8 = ci8 // 8
    kp = pass_chunks(nc8) * (-(-nc8 // pass_chunks(nc8))) * 8
    mt = -(-cout // 16)
    g128, g64 = -(-mt // 8), -(-mt // 4)
    bn = 128 if g128 * 8 <= g64 * 4 else 64
    return f"conv_gemm/{bn}", kp, bn * (-(-mt * 16 // bn))


def report(alpha, h=208, w=416):
    rows, tot = [], [0.0, 0.0, 0.0, 0.0]
    for name, ks, cin, cout, lv in layers(alpha):
        px = (h >> lv) * (w >> lv)
        macs = px * ks * ks * cin * cout
        fam, kp, mp = conv_tile(cin, cout)
        fam_b, kpb, mpb = conv_tile(cout, cin)              # dgrad: the roles of cin / cout swap
        fwd_pad = px * ks * ks * kp * mp + px * ks * ks * kpb * mpb
        wg_pad = px * ks * ks * (64 * -(-pad8(cin) // 64)) * (64 * -(-pad8(cout) // 64)) if max(cin, cout) > 32 else px * ks * ks * pad8(cin) * (16 * -(-cout // 16))
        rows.append((name, ks, cin, cout, fam, kp, mp, 2 * macs / fwd_pad, macs / wg_pad, 3 * macs))
        tot[0] += 2 * macs; tot[1] += fwd_pad; tot[2] += macs; tot[3] += wg_pad
    print(f"alpha {alpha:g}: forward + dgrad logical / padded MACs {tot[0] / tot[1]:.3f}, weight gradient {tot[2] / tot[3]:.3f}, "
          f"all three {(tot[0] + tot[2]) / (tot[1] + tot[3]):.3f}; padded GMAC per image {(tot[1] + tot[3]) / 1e9:.2f} (logical {(tot[0] + tot[2]) / 1e9:.2f})")
    print("   layer    k   cin  cout  kernel          K pad  M pad  fwd+dgrad  wgrad   share of the step's logical MACs")
    all_m = sum(r[9] for r in rows)
    for r in rows:
        print(f"   {r[0]:7s} {r[1]:2d} {r[2]:5d} {r[3]:5d}  {r[4]:14s} {r[5]:5d} {r[6]:6d}   {r[7]:6.3f}   {r[8]:6.3f}   {r[9] / all_m:6.3f}")


if __name__ == "__main__":
    for a in ([float(v) for v in sys.argv[1:]] or [1, 1.25, 1.5, 1.75, 2]):
        report(a)
