"""world_size-2 and world_size-8 gloo checks of the multi-process host logic (runs on CPU): file sharding + the mean-IM-size
reduction give the single-process answer; the gradient averaging path gives the mean of the ranks' gradients; at 8 ranks with
the uneven shards 2335 / 8 produces and a directory smaller than the world."""
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


WORKER8 = textwrap.dedent("""
    import os, sys, json
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    world = int(sys.argv[4])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=world)
    rank = dist.get_rank()
    from inconsistencymasks_amd import functions as F
    U, LAB = 2335, 259                                  # the ISIC-shaped set of BASELINE configs[1] (SURVEY 8)
    names = [f"ISIC_{i:07d}.png" for i in np.random.default_rng(1).permutation(U)]
    size_of = lambda n: int(n[5:12]) * 7919 % 4001      # fake per-image IM size
    pred_of = lambda n: int(n[5:12]) * 104729 % 6007    # fake prediction size
    mine = F.shard_list(names)
    # (1) mean IM size of the directory (functions.py:2889): one tiny reduction of (sum, count)
    tot, cnt = F._all_reduce_sum([sum(size_of(n) for n in mine), len(mine)])
    mean = round(tot / cnt, 0)
    # (2) the keep rule on the shard and the step count the ranks agree on (bench.py / fit: the smallest shard decides)
    kept = [n for n in mine if pred_of(n) > size_of(n) and pred_of(n) > 0]
    lab_mine = F.shard_list([f"lab_{i:04d}.png" for i in range(LAB)])
    cap = torch.tensor([(len(kept) + len(lab_mine)) // 32])
    dist.all_reduce(cap, op=dist.ReduceOp.MIN)
    # (3) rank 0's verdict reaches every rank (bench.py's sharding re-check: broadcast(ok), then everybody leaves together)
    ok = torch.ones(1)
    if rank == 0:
        ok.fill_(0.0 if world == 8 and os.environ.get("IMK_TEST_FAIL_CHECK") == "1" else 1.0)
    dist.broadcast(ok, 0)
    # (4) gradient bucket: mean over ranks, stats ride along
    class M: pass
    m = M()
    m.grads_and_stats = torch.cat([torch.full((5,), float(rank + 1)), torch.tensor([float(rank), 1.0 if rank == 5 else 0.0, 1.0, 0.0])])
    m.grads, m.stats = m.grads_and_stats[:5], m.grads_and_stats[5:]
    scale = F._grad_allreduce(m)
    # (5) sharded benchmark lists with uneven shards
    whole, = F._gather_lists([float(int(n[5:12])) for n in mine])
    # (6) moving statistics averaged
    class P: n_trainable = 2
    m.params = torch.tensor([9.0, 9.0, float(rank), 10.0 * rank])
    m.plan = P()
    F._sync_moving_stats(m)
    # (7) a directory smaller than the world: inference shards may be empty, a TRAINING list is replicated instead
    tiny = [f"t_{i}.png" for i in range(5)]
    tiny_mine = F.shard_list(tiny)
    tiny_train = F._train_shard(tiny)
    tot5, cnt5 = F._all_reduce_sum([sum(int(n[2]) for n in tiny_mine), len(tiny_mine)])
    print(json.dumps({"rank": rank, "n": len(mine), "first": mine[0], "last": mine[-1], "mean": mean, "kept": len(kept), "cap": int(cap),
                      "ok": float(ok), "g": (m.grads * scale).tolist(), "stats": m.stats.tolist(), "whole_n": len(whole),
                      "whole_sorted": whole == sorted(whole), "whole_sum": sum(whole), "params": m.params.tolist(),
                      "tiny": tiny_mine, "tiny_train": tiny_train, "tiny_tot": [tot5, cnt5]}))
    dist.destroy_process_group()
""")


def test_gloo_world8_uneven_shards(tmp_path):
    """the host logic at world 8 with 2 335 files (shards of 291 / 292) and a 5-file directory."""
    import json
    import numpy as np
    script = tmp_path / "w8.py"
    script.write_text(WORKER8)
    port = str(31000 + os.getpid() % 2000)
    world = 8
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(world)], stdout=subprocess.PIPE, text=True,
                              env={**os.environ, "IMK_DIST_CPU": "1", "OMP_NUM_THREADS": "1"}) for r in range(world)]
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    U = 2335
    names = sorted(f"ISIC_{i:07d}.png" for i in np.random.default_rng(1).permutation(U))
    size_of = lambda n: int(n[5:12]) * 7919 % 4001
    pred_of = lambda n: int(n[5:12]) * 104729 % 6007
    assert [o["n"] for o in outs] == [(U * (r + 1)) // 8 - (U * r) // 8 for r in range(8)] and sum(o["n"] for o in outs) == U
    assert set(o["n"] for o in outs) == {291, 292}
    # contiguous blocks of the sorted list, in rank order, no overlap
    pos = 0
    for o in outs:
        assert o["first"] == names[pos] and o["last"] == names[pos + o["n"] - 1]
        pos += o["n"]
    expect_mean = round(sum(size_of(n) for n in names) / U, 0)
    kept_total = sum(1 for n in names if pred_of(n) > size_of(n) and pred_of(n) > 0)
    assert sum(o["kept"] for o in outs) == kept_total
    lab = [(259 * (r + 1)) // 8 - (259 * r) // 8 for r in range(8)]
    expect_cap = min((o["kept"] + l) // 32 for o, l in zip(outs, lab))
    for o in outs:
        assert o["mean"] == expect_mean
        assert o["cap"] == expect_cap                       # every rank runs the same number of all-reduces
        assert o["ok"] == 1.0
        assert o["g"] == [4.5] * 5                          # mean of 1..8
        assert o["stats"][0] == 28.0 and o["stats"][1] == 1.0 and o["stats"][2] == 8.0      # summed: loss terms, ONE rank's overflow flag
        assert o["whole_n"] == U and o["whole_sorted"] and o["whole_sum"] == float(sum(range(U)))
        assert o["params"] == [9.0, 9.0, 3.5, 35.0]
        assert o["tiny_train"] == [f"t_{i}.png" for i in range(5)]
        assert o["tiny_tot"] == [10, 5]
    assert sorted(n for o in outs for n in o["tiny"]) == [f"t_{i}.png" for i in range(5)]
    assert sum(1 for o in outs if not o["tiny"]) == 3       # three ranks have nothing to infer and still take part in the reduction


WORKER_CAND = textwrap.dedent("""
    import os, sys, json
    import torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    world = int(sys.argv[4])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=world)
    from inconsistencymasks_amd import functions as F, im_driver as D
    seen = []
    class M:
        grads_and_stats = torch.ones(4)
    def train_candidate(i, side_by_side=False):
        # what a trainer sees inside: a one-rank run (no sharding, no collective), although the process group exists
        files = [f"f{j}.png" for j in range(10)]
        seen.append({"i": i, "dist_none": F._dist() is None, "rank_world": list(F._rank_world()), "shard": len(F._train_shard(files)),
                     "bench": F.shard_list(files) == sorted(files), "scale": F._grad_allreduce(M())})
        return (f"model_{i}", float(i) / 10, dist.get_rank())
    rows = D.train_candidates([0, 1, 2, 3, 4], train_candidate, world, parallel=1)
    outside = {"dist_none": F._dist() is None, "rank_world": list(F._rank_world())}
    print(json.dumps({"rank": dist.get_rank(), "rows": rows, "seen": seen, "outside": outside,
                      "steps": [D.epoch_steps(2594, 32, world), D.epoch_steps(10, 32, world)]}))
    dist.destroy_process_group()
""")


def _run_cand(tmp_path, world, mode, port_base):
    import json
    script = tmp_path / f"c{world}{mode}.py"
    script.write_text(WORKER_CAND)
    port = str(port_base + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(world)], stdout=subprocess.PIPE, text=True,
                              env={**os.environ, "IMK_DIST_CPU": "1", "OMP_NUM_THREADS": "1", "IM_DP_MODE": mode}) for r in range(world)]
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return sorted(outs, key=lambda o: o["rank"])


def test_whole_candidates_per_rank_host_logic(tmp_path):
    """IM_DP_MODE=candidates (im_driver.train_candidates; SURVEY 8e row 3, ISIC_2018/09_ISIC_2018_IM.py:90): candidate at position p of
    the list trains on rank p mod N inside functions.local_rank_scope -- where the package behaves as ONE rank -- and every rank gets
    all rows, in candidate order; epoch steps are the reference's (files // 32), not files // (32 N).  World 2 and 8, gloo, CPU."""
    for world, port in ((2, 33000), (8, 35000)):
        outs = _run_cand(tmp_path, world, "candidates", port)
        for o in outs:
            assert o["rows"] == [[f"model_{i}", i / 10, i % world] for i in range(5)]           # gathered, ordered, trained on i mod N
            assert [s["i"] for s in o["seen"]] == [i for i in range(5) if i % world == o["rank"]]
            for s in o["seen"]:
                assert s["dist_none"] and s["rank_world"] == [0, 1] and s["shard"] == 10 and s["bench"] and s["scale"] == 1.0
            assert o["outside"] == {"dist_none": False, "rank_world": [o["rank"], world]}         # the scope ends with the candidate
            assert o["steps"] == [81, 1]
    # the default mode keeps the data-parallel contract: every rank trains every candidate, steps = files // (32 N)
    outs = _run_cand(tmp_path, 2, "gradient", 37000)
    for o in outs:
        assert [s["i"] for s in o["seen"]] == [0, 1, 2, 3, 4] and not o["seen"][0]["dist_none"] and o["seen"][0]["scale"] == 0.5
        assert o["steps"] == [40, 1]
