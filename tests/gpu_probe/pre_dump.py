"""GPU: where does the decoder's first-stage launch (conv_pipe_kernel<..., PRE>) differ from its two launches?
  build (here):  python tests/gpu_probe/pre_dump.py build      -> build/predbg/libimk_predbg.so (-DIMK_PRE_DEBUG)
  run (GPU box): IMK_LIB_PATH=build/predbg/libimk_predbg.so python tests/gpu_probe/pre_dump.py
The probe build makes the PRE kernel write, for every pixel, the first stage's pre-BatchNorm output (what d9.ca's own launch stores)
and the BatchNorm'd value it hands to the 3x3.  Compared per pixel with the materialized two-launch path of the same model."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "predbg", "libimk_predbg.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from inconsistencymasks_amd import build as B
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs, procs = [], []
    for s in sorted(glob.glob(os.path.join(B.CSRC, "*.hip"))):
        o = os.path.join(os.path.dirname(OUT), os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + ["-DIMK_PRE_DEBUG", "-c", s, "-o", o]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    print(OUT)
    sys.exit(0)

import numpy as np, torch
from inconsistencymasks_amd.unet import UNet
h, w, c, k, alpha = 64, 80, 3, 1, 0.5
tot = 0
for seed in (11, 12, 13, 14, 15, 16):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randint(0, 256, (6, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
    y = (torch.rand((6, h, w, k), device="cuda", generator=g) > 0.6).to(torch.uint8)
    m = UNet(h, w, c, k, alpha, "sigmoid", seed=seed)
    for _ in range(3):
        m.train_step(x, y, 0, 3e-3, 1e-4)
    m.predict_device(x)                                   # PRE path: dumps in d9.c3 (post-BN) / d9.ca (pre-BN)
    pre_z, pre_w, pre_last = (m.intermediate(n, 6, 0).numpy().copy() for n in ("d9.ca", "d9.c3", "d9.c1"))
    block_out = ["e1.c1", "e2.c1", "e3.c1", "e4.c1", "b.c1", "d6.c1", "d7.c1", "d8.c1", "d9.c1"]      # stored by both paths
    fused = {n: m.intermediate(n, 6, 0).numpy().copy() for n in block_out}
    m.debug(materialize=True)
    m.predict_device(x)                                   # two launches, everything stored
    ref_z, ref_last = m.intermediate("d9.ca", 6, 0).numpy().copy(), m.intermediate("d9.c1", 6, 0).numpy().copy()
    first = [(n, int((fused[n] != m.intermediate(n, 6, 0).numpy()).sum())) for n in block_out]
    print("   block outputs differing (fused inference vs materialized):", first)
    m.debug(materialize=False)
    sd = m.state_dict()
    l = [q for q in m.plan.layers if q["name"] == "d9.bna"][0]
    # folded inference scale / shift as the library computes them (imk_elem.hip: bn_fold): sc = gamma / sqrt(var + eps), sh = beta - mean sc
    gam, bet, mu, var = (sd["d9.bna." + n].double().numpy() for n in ("gamma", "beta", "mean", "var"))
    dz = pre_z != ref_z
    print(f"seed {seed}: first stage pre-BN differs at {int(dz.sum())} of {dz.size} values; last block output differs at {int((pre_last != ref_last).sum())}")
    idx = np.argwhere(pre_last != ref_last)
    for (b, yy, xx, ch) in idx[:6]:
        print(f"   d9.c1[{b},{yy},{xx},{ch}] pre {pre_last[b, yy, xx, ch]:.6f} two-launch {ref_last[b, yy, xx, ch]:.6f}; first stage pre-BN equal in the 3x3 window: "
              f"{bool((pre_z[b, max(yy-1,0):yy+2, max(xx-1,0):xx+2] == ref_z[b, max(yy-1,0):yy+2, max(xx-1,0):xx+2]).all())}")
    tot += int((pre_last != ref_last).sum())
    np.savez(os.path.join(ROOT, "gpurun_out", f"pre_dump_{seed}.npz"), pre_z=pre_z, pre_w=pre_w, ref_z=ref_z, pre_last=pre_last, ref_last=ref_last,
             gamma=gam, beta=bet, mean=mu, var=var)
print("total differing outputs", tot)
