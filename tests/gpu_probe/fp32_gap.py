"""GPU probabilities against the INDEPENDENT oracle (unet_oracle.forward(emulate_fp16=False): plain fp32, no knowledge of the kernels'
rounding points), every parity configuration of tests/test_gpu_unet.py, default-init and randomised-BatchNorm weights: rel-L2, max |dp|,
decision-flip rate -- and the same three numbers for the fp16-emulating oracle against the fp32 one (what fp16 storage costs on its own).
The bounds of test_inference_parity / test_baseline_shapes_full_size (check_against_fp32) are frozen from this table:
tests/measured/fp32_gap_r06.json is its last line.
    python tests/gpu_probe/fp32_gap.py > gpurun_out/fp32_gap.txt"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_unet as T          # noqa: E402
from inconsistencymasks_amd.unet import UNet  # noqa: E402
from oracle import unet_oracle as U           # noqa: E402


def three(p, r, act):
    a, b, c = T.three(p, r, act)
    return [round(a, 6), round(b, 5), round(c, 6)]


out = {}
for group, cfgs, seeds in (("CFGS", T.CFGS, (1, 2, 3)), ("BASELINE_SHAPES", T.BASELINE_SHAPES, (61, 62, 63))):
    for name, cfg in cfgs.items():
        c, k, alpha, act = cfg["c"], cfg["k"], cfg["alpha"], cfg["act"]
        x, _, _ = T.make_input(cfg, seeds[2])
        xd = torch.from_numpy(x).cuda()
        for kind in ("default_init", "random_bn"):
            m = UNet(cfg["h"], cfg["w"], c, k, alpha, act, seed=seeds[0])
            sd = m.state_dict()
            if kind == "random_bn":
                sd = T.randomize_bn(sd, seeds[1])
                m.load_state_dict(sd)
            p = m.predict_device(xd).cpu().numpy()
            r32 = U.forward(sd, x, c, k, alpha, act, emulate_fp16=False).numpy()
            r16 = U.forward(sd, x, c, k, alpha, act, emulate_fp16=True).numpy()
            out[f"{group}.{name}.{kind}"] = {"gpu_vs_fp32": three(p, r32, act), "fp16emu_vs_fp32": three(r16, r32, act),
                                             "gpu_vs_fp16emu": three(p, r16, act)}
            print(f"{group}.{name}.{kind}", out[f"{group}.{name}.{kind}"], flush=True)
print(json.dumps(out))
