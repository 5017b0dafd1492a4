"""GPU: does the data-parallel recipe still train?  

bench.py / the drivers scale by REPLICAS: per-GPU batch 32, global batch 32 N, epoch steps = images // (32 N), unchanged
lr = 3e-3 / wd = 1e-4 (config.ini:10-11), BatchNorm batch statistics per GPU, moving statistics averaged over the replicas
at the end of every epoch.  This probe emulates N ranks in ONE process exactly: per optimizer step N micro-batches of 32, each
with its own BatchNorm batch statistics AND its own copy of the moving statistics (swapped in and out of the parameter
vector's tail), gradients averaged, one AdamW update; at the end of every epoch the N moving-statistics copies are averaged --
then val IoU / dice of the synthetic ISIC task over the reference's 50 epochs (config.ini:3).
    python tests/gpu_probe/dp_convergence.py  ->  gpurun_out/r3_dp_convergence.txt
WORLDS=1,8 SEEDS=0,1,2 EPOCHS=50 override."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from inconsistencymasks_amd.unet import UNet

cfg = dict(bench.CONFIGS["isic"])
H, W, C, K, ALPHA = cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"]
dev = torch.device("cuda")
worlds = [int(v) for v in os.environ.get("WORLDS", "1,2,4,8").split(",")]
seeds = [int(v) for v in os.environ.get("SEEDS", "0,1,2").split(",")]
EPOCHS = int(os.environ.get("EPOCHS", 50))
LR_RULE = os.environ.get("LR_RULE", "none")          # none | sqrt | linear: lr x 1 / sqrt(world) / world (the reference has one GPU: no rule)
OUT = os.environ.get("OUT", "r3_dp_convergence.txt")
BN_RULE = os.environ.get("BN_RULE", "none")          # none | scaled: BatchNorm moving-statistics momentum 0.99 / 0.99 ** world
N_TRAIN = cfg["unlabeled"] + cfg["labeled"]            # 2594: the size of a generation's training directory
x, m = bench.synth_images(cfg, N_TRAIN, 42, dev)
y = (m // 255).contiguous()
xv, mv = bench.synth_images(cfg, 100, 4343, dev)
gv = mv[..., 0] > 0
show = [e for e in (1, 2, 3, 5, 10, 20, 30, 40, 50) if e <= EPOCHS]


def val_metrics(model):
    model.repack()
    p = torch.cat([model.predict_device(xv[i:i + 50]) for i in range(0, 100, 50)])[..., 0] > 0.5
    inter = (p & gv).sum(dim=(1, 2)).double()
    union = (p | gv).sum(dim=(1, 2)).double()
    tot = p.sum(dim=(1, 2)).double() + gv.sum(dim=(1, 2)).double()
    return float((inter / union.clamp(min=1)).mean()), float((2 * inter / tot.clamp(min=1)).mean())


def run(world, seed):
    model = UNet(H, W, C, K, ALPHA, "sigmoid", seed=100 + seed, device=dev)
    model.init_train_state()
    if BN_RULE == "scaled":
        model.set_bn_momentum(0.99 ** world)
    nt = model.plan.n_trainable
    mov = [model.params[nt:].clone() for _ in range(world)]          # each rank's BatchNorm moving statistics
    gen = torch.Generator(device=dev).manual_seed(seed)
    steps = N_TRAIN // (32 * world)
    curve = {}
    for ep in range(1, EPOCHS + 1):
        perm = torch.randperm(N_TRAIN, device=dev, generator=gen)
        for s in range(steps):
            acc = torch.zeros_like(model.grads_and_stats)
            for r in range(world):
                # rank r's block of the epoch's shuffle (contiguous blocks of the permuted set, as shard_list cuts the sorted one)
                idx = perm[(r * steps + s) * 32:(r * steps + s + 1) * 32]
                model.params[nt:] = mov[r]
                model.fwd_bwd(x[idx].contiguous(), y[idx].contiguous(), 0)
                acc += model.grads_and_stats                           # = the all-reduce(sum) of the flat bucket
                mov[r] = model.params[nt:].clone()
            model.grads_and_stats.copy_(acc)
            lr = bench.LR * {"none": 1.0, "sqrt": world ** 0.5, "linear": float(world)}[LR_RULE]
            model.adamw_step(lr, bench.WD, grad_scale=1.0 / world)
        avg = torch.stack(mov).mean(0)                                  # functions._sync_moving_stats at every epoch end
        mov = [avg.clone() for _ in range(world)]
        model.params[nt:] = avg
        if ep in show:
            curve[ep] = val_metrics(model)
    return curve


lines = ["Data-parallel recipe, emulated exactly in one process (tests/gpu_probe/dp_convergence.py): synthetic ISIC task, %d training"
         " images, 100 validation images," % N_TRAIN,
         "per-rank batch 32, lr 3e-3 (scaling rule with the world size: " + LR_RULE + "), BatchNorm momentum rule: " + BN_RULE + ", wd 1e-4, %d epochs; val IoU / dice (mean over the validation images, threshold 0.5) after epoch e."
         % EPOCHS, "world = number of emulated ranks (global batch 32 x world, %s optimizer steps per epoch)."
         % ", ".join(f"{N_TRAIN // (32 * w)}" for w in worlds), ""]
final = {}
for world in worlds:
    for seed in seeds:
        c = run(world, seed)
        final.setdefault(world, []).append(c[show[-1]])
        lines.append(f"world {world} seed {seed}: " + "  ".join(f"e{e}: {c[e][0]:.4f}/{c[e][1]:.4f}" for e in show))
        print(lines[-1], flush=True)
lines.append("")
for world in worlds:
    ious = [v[0] for v in final[world]]
    dices = [v[1] for v in final[world]]
    lines.append(f"world {world}: final val IoU mean {sum(ious) / len(ious):.4f} (min {min(ious):.4f}, max {max(ious):.4f}), "
                 f"dice mean {sum(dices) / len(dices):.4f}")
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", OUT), "w").write("\n".join(lines) + "\n")
