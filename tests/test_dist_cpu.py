"""world_size-2 gloo checks of the multi-process host logic (runs on CPU): file sharding + the mean-IM-size
reduction give the single-process answer; the gradient averaging path gives the mean of the ranks' gradients."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
    from inconsistencymasks_amd import functions as F
    names = [f"img_{i:04d}.png" for i in np.random.default_rng(0).permutation(101)]
    size_of = lambda n: int(n[4:8]) * 7 % 1000          # fake per-image IM size
    mine = F.shard_list(names)
    tot, cnt = F._all_reduce_sum([sum(size_of(n) for n in mine), len(mine)])
    mean = round(tot / cnt, 0)
    class M: pass
    m = M()
    m.grads_and_stats = torch.cat([torch.full((5,), float(dist.get_rank() + 1)),          # the model's one flat bucket:
                                   torch.tensor([float(dist.get_rank()), 0.0, 1.0, 0.0])])   # gradients | step statistics
    m.grads, m.stats = m.grads_and_stats[:5], m.grads_and_stats[5:]
    scale = F._grad_allreduce(m)
    # sharded benchmark: per-image metric lists of the ranks -> the complete list, in sorted file order, on every rank
    part = [float(int(n[4:8])) for n in mine]
    whole, = F._gather_lists(part)
    # BatchNorm moving statistics: averaged over the replicas before a checkpoint decision
    class P: n_trainable = 3
    m.params = torch.tensor([9.0, 9.0, 9.0, float(dist.get_rank()), 2.0 + 2 * dist.get_rank()])
    m.plan = P()
    F._sync_moving_stats(m)
    # fewer training files than ranks: every rank keeps the whole list
    few = F._train_shard(["only.png"])
    # an UNSEEDED get_unet() (what the reference's scripts call): each rank's global RNG differs, the replicas must not
    from inconsistencymasks_amd.unet import UNet
    torch.manual_seed(100 + dist.get_rank())
    u = UNet(32, 32, 3, 1, 0.5, "sigmoid", seed=None, device="cpu")
    print(json.dumps({"rank": dist.get_rank(), "n": len(mine), "mean": mean, "g": (m.grads * scale).tolist(),
                      "stats": m.stats.tolist(), "whole": whole, "params": m.params.tolist(), "few": few,
                      "unet_seed": u.seed, "unet_sum": float(u.params.double().abs().sum())}))
    dist.destroy_process_group()
""")


def test_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = str(29000 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, text=True,
                              env={**os.environ, "IMK_DIST_CPU": "1"}) for r in range(2)]
    import json
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=180)
        assert p.returncode == 0, o
        outs.append(json.loads(o.strip().splitlines()[-1]))
    import numpy as np
    names = [f"img_{i:04d}.png" for i in np.random.default_rng(0).permutation(101)]
    expect = round(sum(int(n[4:8]) * 7 % 1000 for n in names) / len(names), 0)
    assert sorted(o["n"] for o in outs) == [50, 51]
    for o in outs:
        assert o["mean"] == expect
        assert o["g"] == [1.5] * 5                      # mean of rank gradients 1 and 2
        assert o["stats"][0] == 1.0                     # summed ride-along stats
        assert o["whole"] == [float(int(n[4:8])) for n in sorted(names)]
        assert o["params"] == [9.0, 9.0, 9.0, 0.5, 3.0]  # trainable part untouched, moving statistics averaged
        assert o["few"] == ["only.png"]
    assert outs[0]["unet_seed"] == outs[1]["unet_seed"] and outs[0]["unet_sum"] == outs[1]["unet_sum"]   # rank 0's draw on every rank
