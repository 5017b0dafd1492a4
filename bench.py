#!/usr/bin/env python3
"""Benchmark of the north-star metric: images/sec per IM generation (ensemble inference + IM creation +
one U-Net training epoch) on ISIC-2018-shaped synthetic data, 256x256x3, 2-model ensemble, alpha = 0.5.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the ranks read
RANK / LOCAL_RANK / WORLD_SIZE), or started plainly -- then this process spawns exactly that launcher as a child BEFORE
it touches the GPU, waits, and exits with the child's code.  `n_gpus` in the output is dist.get_world_size(); a world
size different from --gpus is an error.

One "step" = one IM generation:
  1. N-model ensemble forward over the unlabeled images (batches of --infer-batch) fused with the IM chain
     (threshold -> agreement -> IM -> blocking -> sizes)                       functions.py:2844-2887
  2. keep rule pred_size > im_size and pred_size > 0                           functions.py:2878-2886
  3. one epoch (batch 32 per GPU) of a fresh U-Net on kept pseudo-labels + labelled set,
     mse loss, tfa-AdamW(3e-3, 1e-4)                                            functions.py:207-218
Inputs are resident in HBM when the timed region starts (PNG decode/encode is excluded on both the GPU and
the CPU side).

Multi-GPU, default `--scaling strong` (the north star: "shard the unlabeled image set across the GPUs"): ONE
2335-image set; rank r runs inference + IM on its contiguous block of the sorted set (functions.shard_list, no
collective) and trains on its own kept pairs + its block of the labelled set with batch 32 per GPU, one all-reduce of
the flat fp32 gradient buffer per step (RCCL), epoch steps = images // (32 * N).  value = 2335 / time per generation.
After the timed region rank 0 recomputes the IM stage over the WHOLE set and checks that the ranks' summed IM size,
prediction size and kept count equal it (sharding does not change a single mask).
`--scaling weak`: every rank owns a full-size set (value = N * 2335 / time).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 256
C = 3
K = 1
ALPHA = 0.5
N_MODELS = 2
U_UNLABELED = 2335      # ISIC-2018 Task-1: 2594 train images, 10/90 split
U_LABELED = 259
BATCH = 32
LR, WD = 3e-3, 1e-4
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def synth_images(n, seed, device):
    """Seeded ISIC-like images: smooth low-frequency field + elliptical 'lesion' + noise; mask = the ellipse."""
    g = torch.Generator(device=device).manual_seed(seed)
    yy = torch.arange(H, device=device, dtype=torch.float32)[None, :, None]
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, None, :]
    r = lambda *s: torch.rand(*s, device=device, generator=g)
    field = torch.zeros((n, H, W), device=device)
    for _ in range(4):
        fy, fx, ph = r(n, 1, 1) * 0.05, r(n, 1, 1) * 0.05, r(n, 1, 1) * 6.28
        field += torch.cos(yy * fy + xx * fx + ph)
    cy, cx = (0.3 + 0.4 * r(n, 1, 1)) * H, (0.3 + 0.4 * r(n, 1, 1)) * W
    ry, rx = (0.12 + 0.2 * r(n, 1, 1)) * H, (0.12 + 0.2 * r(n, 1, 1)) * W
    ell = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) < 1
    base = 150 + 15 * field - ell * (40 + 50 * r(n, 1, 1))
    img = base[..., None] + torch.tensor([10.0, -5.0, -15.0], device=device) + (r(n, H, W, C) * 16 - 8)
    return img.clamp(0, 255).to(torch.uint8).contiguous(), (ell.to(torch.uint8) * 255)[..., None].contiguous()


def _usable_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def _calibrate_threads(candidates, run, warm):
    """fastest thread count for `run` (tiny batch-1 convolutions get SLOWER with hundreds of threads; a batch-32 training
    step wants many more): stops once a setting is 1.5x slower than the best so far"""
    best = None
    tried = {}
    for nt in candidates:
        torch.set_num_threads(nt)
        warm()
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        tried[nt] = round(dt, 4)
        if best is None or dt < best[1]:
            best = (nt, dt)
        if dt > 1.5 * best[1]:
            break
    return best[0], tried


def parity_sample(models, x_dev, n=4):
    """"also compare outputs" -- the bench's own ensemble on a few of its images, GPU against the oracle
    (fp16-emulating torch-CPU restatement): largest probability difference, decision flips, and the IM pixels that differ
    when the oracle's IM chain runs on the oracle's probabilities (the IM chain itself is bit-exact given equal inputs)."""
    from oracle import im_oracle, unet_oracle as U
    x = x_dev[:n].cpu().numpy()
    gpu = [m.predict_device(x_dev[:n]).cpu().numpy() for m in models]
    ref = [U.forward(m.state_dict(), x, C, K, ALPHA, "sigmoid", emulate_fp16=True).numpy() for m in models]
    dp = max(float(np.abs(g - r).max()) for g, r in zip(gpu, ref))
    flips = float(np.mean([((g > 0.5) != (r > 0.5)).mean() for g, r in zip(gpu, ref)]))
    im_diff = 0
    for i in range(n):
        a = im_oracle.im_binary(np.stack([g[i] for g in gpu], 0), 0.5, False)["im"]
        b = im_oracle.im_binary(np.stack([r[i] for r in ref], 0), 0.5, False)["im"]
        im_diff += int((a != b).sum())
    return {"images": n, "max_abs_dp": round(dp, 5), "decision_flip_rate": round(flips, 6),
            "im_pixels_differing": im_diff, "im_pixels_total": n * H * W,
            "note": "GPU vs fp16-emulating oracle on the bench's trained ensemble; tolerance of the parity tests: |dp| <= 3e-2, flips <= 1 %"}


def cpu_baseline():
    """The oracle (torch-CPU fp32 restatement, reference-structured: batch-1 forward per image per model, numpy IM,
    batch-32 training) on a bounded sample, extrapolated to one generation.  BASELINE.md section 3: >= 3 repetitions,
    median; thread counts calibrated SEPARATELY for the batch-1 forwards and for the batch-32 training step (up to every
    usable core), both reported."""
    from oracle import im_oracle, unet_oracle as U
    rng = np.random.default_rng(0)
    n_img, reps = 32, 3
    x = rng.integers(0, 256, (max(n_img, BATCH), H, W, C)).astype(np.uint8)
    models = [U.init_weights(C, K, ALPHA, 1000 + j) for j in range(N_MODELS)]
    ncpu = _usable_cpus()
    cand = lambda lo: [n for n in (1, 2, 4, 8, 16, 32, 64, 128, 256) if lo <= n <= ncpu] or [1]
    fwd = lambda k: U.predict_batch1(models[0], x[:k], C, K, ALPHA, "sigmoid")
    nt_fwd, tried_fwd = _calibrate_threads(cand(4), lambda: fwd(4), lambda: fwd(1))
    torch.set_num_threads(nt_fwd)
    t_inf_reps, t_im_reps, t_inf_b_reps = [], [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        preds = [U.predict_batch1(m, x[:n_img], C, K, ALPHA, "sigmoid") for m in models]
        t_inf_reps.append((time.perf_counter() - t0) / n_img)               # per image, all N models
        t0 = time.perf_counter()
        for i in range(n_img):
            r = im_oracle.im_binary(np.stack([p[i] for p in preds], 0), 0.5, False)
            im_oracle.block(x[i], [r["final"][0]], r["im"], True, True)
        t_im_reps.append((time.perf_counter() - t0) / n_img)
    t_inf, t_im = _median(t_inf_reps), _median(t_im_reps)
    # the training step: its own thread count
    p = U.init_weights(C, K, ALPHA, 7)
    opt = U.new_opt_state(p)
    y = (rng.random((BATCH, H, W, K)) > 0.5).astype(np.float32)
    step = lambda: U.train_step(p, opt, x[:BATCH], y, C, K, ALPHA, "sigmoid", "mse")
    step()                                                                   # one-off autograd warm-up
    nt_train, tried_train = _calibrate_threads(cand(8), step, lambda: None)
    torch.set_num_threads(nt_train)
    t_step_reps = []
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        t_step_reps.append(time.perf_counter() - t0)
    t_step = _median(t_step_reps)
    # the same forwards as ONE batch per model (so that the ratio is not inflated by the reference's
    # batch-1 choice), at the training step's thread count; reported beside the reference-structured headline
    t0 = time.perf_counter()
    with torch.no_grad():
        for m in models:
            U.forward(m, x[:n_img], C, K, ALPHA, "sigmoid")
    t_inf_batched = (time.perf_counter() - t0) / n_img
    steps = (U_UNLABELED + U_LABELED) // BATCH
    t_gen = U_UNLABELED * (t_inf + t_im) + steps * t_step
    return {"value": round(U_UNLABELED / t_gen, 3), "unit": "images/s", "cores": max(nt_fwd, nt_train), "kind": "port",
            "sample": f"median of {reps} repetitions of: {n_img} images x {N_MODELS} models batch-1 fp32 forward + numpy IM "
                      f"({nt_fwd} threads), 1 train step of batch {BATCH} ({nt_train} threads); extrapolated to "
                      f"U={U_UNLABELED}, {steps} steps; {ncpu} CPUs usable",
            "threads_forward": nt_fwd, "threads_train_step": nt_train,
            "thread_calibration_s": {"forward_4_images": tried_fwd, "train_step": tried_train},
            "t_infer_per_image_s": round(t_inf, 5), "t_im_per_image_s": round(t_im, 6), "t_train_step_s": round(t_step, 4),
            "repetitions": {"t_infer_per_image_s": [round(v, 5) for v in t_inf_reps],
                            "t_train_step_s": [round(v, 4) for v in t_step_reps]},
            "batched_variant": {"t_infer_per_image_s": round(t_inf_batched, 5),
                                "value": round(U_UNLABELED / (U_UNLABELED * (t_inf_batched + t_im) + steps * t_step), 3),
                                "note": f"forwards as one batch of {n_img} per model instead of batch 1, {nt_train} threads"}}


def png_io_rate(images):
    """PNG encode / decode rate of the host path the directory API uses (Pillow on a thread pool; SURVEY H5: excluded
    from the headline on both the GPU and the CPU side, reported separately)."""
    import io
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    threads = int(os.environ.get("IMK_IO_THREADS", min(16, os.cpu_count() or 8)))
    def enc(a):
        b = io.BytesIO()
        Image.fromarray(a).save(b, format="PNG", compress_level=1)
        return b.getvalue()
    def dec(b):
        return np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
    with ThreadPoolExecutor(max_workers=threads) as pool:
        t0 = time.perf_counter(); blobs = list(pool.map(enc, images)); t1 = time.perf_counter()
        back = list(pool.map(dec, blobs)); t2 = time.perf_counter()
    assert np.array_equal(back[0], images[0])
    return {"encode_images_per_s": round(len(images) / (t1 - t0), 1), "decode_images_per_s": round(len(images) / (t2 - t1), 1),
            "threads": threads, "sample": f"{len(images)} images 256x256x3, Pillow compress_level=1"}


# minimum HBM bytes of one forward pass with one kernel per block and fp16 activations (ISIC, alpha 0.5):
# every tensor that crosses a block boundary written once and read once + the uint8 input + the fp32 output
FWD_MIN_BYTES_PER_IMAGE = 10_551_296
IM_BYTES_PER_IMAGE = 1_048_576          # probability stack + image in, image + mask + IM out

FAMILIES = ["conv_mfma_kernel<16,1>", "conv_mfma_kernel<16,2>", "conv_mfma_kernel<16,4>", "conv_mfma_kernel<8,1>",
            "conv_mfma_kernel<8,2>", "conv_mfma_kernel<8,4>", "conv_pipe_kernel", "wgrad_mfma_kernel", "bn_bwd_prep_kernel",
            "bn_bwd_coef_kernel", "bn_finalize_kernel", "wgf_stage1+2_kernel", "head_kernel", "head_loss_kernel",
            "step_tail(loss_finalize,adamw,pack,fold)", "im_kernel"]


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start `python -m torch.distributed.run` as a CHILD (never exec: nothing in this
    process has touched the GPU yet, and it must stay that way) and return its exit code."""
    import socket
    import subprocess
    one_gpu = os.environ.get("IMK_BENCH_ONE_GPU") == "1"
    n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if n_dev < args.gpus and not one_gpu:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--infer-batch", type=int, default=256)
    ap.add_argument("--images", type=int, default=U_UNLABELED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="no per-launch event timing (roofline fields empty)")
    ap.add_argument("--prof-period", type=int, default=61, help="time every k-th hooked kernel launch with HIP events")
    ap.add_argument("--pretrain-steps", type=int, default=300)
    ap.add_argument("--bn-settle-steps", type=int, default=600)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # IMK_BENCH_ONE_GPU=1 + IMK_BENCH_BACKEND=gloo: functional test of the multi-rank path on a single-GPU box
    if os.environ.get("IMK_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IMK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    n_gpus = dist.get_world_size() if world > 1 else 1
    strong = args.scaling == "strong"

    import ctypes
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd import im as imk_im
    from inconsistencymasks_amd._lib import lib as imk_lib
    from inconsistencymasks_amd.unet import UNet

    NV = len(FAMILIES)
    def prof_collect():
        pc = (ctypes.c_int64 * NV)(); pms = (ctypes.c_double * NV)(); pby = (ctypes.c_double * NV)()
        imk_lib.imk_prof_collect(pc, pms, pby)
        return [int(v) for v in pc], [float(v) for v in pms], [float(v) for v in pby]
    # HIP events (on the launch stream) around every k-th hooked launch, from process start: the setup phase is
    # collected separately so that the whole-process average can be compared with `rocprofv3 --stats` of this command
    imk_lib.imk_prof_enable(0 if args.no_prof else args.prof_period)

    # ---- synthetic, HBM-resident inputs ---------------------------------------------------------------------------
    # strong: every rank generates THE set (seed 42) and keeps its contiguous block (functions.shard_list's rule);
    # weak:   every rank owns a full-size set of its own
    U_total = args.images
    if strong:
        x_all, _ = synth_images(U_total, 42, dev)
        xl_all, ml_all = synth_images(U_LABELED, 4242, dev)
        cut = lambda n: ((n * rank) // world, (n * (rank + 1)) // world)
        (u0, u1), (l0, l1) = cut(U_total), cut(U_LABELED)
        x_unl = x_all[u0:u1].contiguous()
        x_lab, m_lab = xl_all[l0:l1].contiguous(), ml_all[l0:l1].contiguous()
        x_pre, y_pre = xl_all, (ml_all // 255).contiguous()         # the ensemble is (pre)trained on the whole labelled set
        if rank != 0:
            del x_all
    else:
        x_unl, _ = synth_images(U_total, 42 + 1000 * rank, dev)
        x_lab, m_lab = synth_images(U_LABELED, 4242 + 1000 * rank, dev)
        x_pre, y_pre = x_lab, (m_lab // 255).contiguous()
    y_lab = (m_lab // 255).contiguous()
    U = x_unl.shape[0]

    # ---- ensemble: seeded he_normal, briefly trained on the labelled set so that predictions are lesions (the same
    # on every rank: replicated weights)
    models = []
    for j in range(N_MODELS):
        m = UNet(H, W, C, K, ALPHA, "sigmoid", seed=1000 + j, device=dev)
        g = torch.Generator(device=dev).manual_seed(j)
        for it in range(args.pretrain_steps + args.bn_settle_steps):
            idx = torch.randint(0, x_pre.shape[0], (BATCH,), device=dev, generator=g)
            # Keras BN momentum 0.99 needs ~500 steps before the moving statistics (what inference uses) have
            # forgotten their initial values; the reference trains 4050 steps.  Settle them with lr = wd = 0.
            lr, wd = (LR, WD) if it < args.pretrain_steps else (0.0, 0.0)
            m.train_step(x_pre[idx].contiguous(), y_pre[idx].contiguous(), 0, lr, wd)
        m.repack()   # fold the moving statistics for inference
        models.append(m)
    ens = F.EnsembleIM(models)
    student = UNet(H, W, C, K, ALPHA, "sigmoid", seed=7, device=dev)
    init_params = student.params.clone()
    gen_perm = torch.Generator(device=dev).manual_seed(42 + rank)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    info = {}

    # Whole-set buffers: the IM kernel writes every batch's blocked images / pseudo-labels straight into [0, U), the
    # labelled pairs sit behind them once -- the "directory" the reference builds (kept pseudo-labelled pairs + labelled
    # pairs) is then ONE gather with the composed (keep, shuffle) index instead of cat / index / cat / index passes.
    pool_x = torch.empty((U + x_lab.shape[0], H, W, C), dtype=torch.uint8, device=dev)
    pool_y = torch.empty((U + x_lab.shape[0], 1, H, W), dtype=torch.uint8, device=dev)
    pool_x[U:] = x_lab
    pool_y[U:, 0] = m_lab[..., 0]                                   # {0, 255} like the written mask files
    lab_idx = torch.arange(U, U + x_lab.shape[0], device=dev)

    def infer_batches(n):
        """batches of --infer-batch images; a last batch under a quarter of that is spread over the others instead (at 8 ranks a
        shard is 292 images: one call, not 256 + 36 -- the deep levels of a forward cost the same for 36 images as for 256)"""
        k, b = -(-n // args.infer_batch), args.infer_batch
        if k > 1 and n - (k - 1) * b < b // 4:
            k -= 1
            b = -(-n // k)
        return [(i, min(i + b, n)) for i in range(0, n, b)]

    def im_stage(x, into_pool=True):
        ps, ims = [], []
        for i, j in infer_batches(x.shape[0]):
            r = ens.run(x[i:j], 0.5, False, True, True, out={"img_out": pool_x[i:j], "masks": pool_y[i:j]} if into_pool else None)
            ps.append(r["pred_size"][:, 0]); ims.append(r["im_size"][:, 0])
        ps, ims = torch.cat(ps), torch.cat(ims)
        return ps, ims, (ps > ims) & (ps > 0)            # keep rule functions.py:2878-2886

    def generation(record=None):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        ps, ims, keep = im_stage(x_unl)
        e1.record()
        # training set = kept pseudo-labelled pairs + labelled pairs (the directory the reference builds)
        src = torch.cat([torch.nonzero(keep).squeeze(1), lab_idx])
        n_train = src.shape[0]
        steps = n_train // BATCH
        if world > 1:   # every rank must run the same number of gradient all-reduces: the smallest shard decides
            st = torch.tensor([steps], device=dev)
            dist.all_reduce(st, op=dist.ReduceOp.MIN)
            steps = int(st.item())
        student.params.copy_(init_params)
        student._packed_ok = False
        student.init_train_state()
        src = src[torch.randperm(n_train, device=dev, generator=gen_perm)]
        tx = pool_x[src]                     # the epoch's shuffle and the keep filter as one gather; batches are contiguous views
        ty = pool_y[src].reshape(n_train, H, W, 1) // 255            # parse_image_ISIC_2018: 255 -> 1 (functions.py:955-977)
        for s in range(steps):
            student.fwd_bwd(tx[s * BATCH:(s + 1) * BATCH], ty[s * BATCH:(s + 1) * BATCH], 0)
            scale = F._grad_allreduce(student)
            student.adamw_step(LR, WD, grad_scale=scale)
        e2.record()
        if record is not None:
            record.append((e0, e1, e2))
        info.update(epoch_steps=int(steps), n_train=n_train)
        return torch.stack([ps.sum(), ims.sum(), keep.sum()]).double()      # read after the timed region

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        generation()
    barrier()
    setup_prof = prof_collect()
    rec = []
    t0 = time.perf_counter()
    # Sampling: an event record is a barrier packet on the stream (~5 us before the next kernel starts), so bracketing
    # every launch costs ~10 % of a training step; every 61st hooked launch (prime, coprime to the ~135 launches of a
    # training step and the 18 of a forward) costs < 1 % and still yields a few hundred samples.
    for k in range(args.steps):
        totals = generation(rec)
    barrier()
    elapsed = time.perf_counter() - t0
    pc, pms, pby = prof_collect()
    imk_lib.imk_prof_enable(0)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    t_inf = sum(a.elapsed_time(b) for a, b, _ in rec) / len(rec)
    t_ep = sum(b.elapsed_time(c) for _, b, c in rec) / len(rec)
    totals = torch.cat([totals, torch.tensor([float(info.pop("n_train"))], dtype=torch.float64, device=dev)])
    if world > 1:
        dist.all_reduce(totals)
    n_all = U_total * (1 if strong else world)
    info.update(kept=int(totals[2]), train_images=int(totals[3]), mean_pred_size=round(float(totals[0]) / n_all, 1),
                mean_im_size=round(float(totals[1]) / n_all, 1))

    # ---- sharding changes nothing: the whole set on one rank gives the ranks' summed sizes and kept count --------------
    sharding_check = None
    if strong and world > 1 and rank == 0:
        ps, ims, keep = im_stage(x_all, into_pool=False)
        whole = [float(ps.sum()), float(ims.sum()), float(keep.sum())]
        sharding_check = {"sum_pred_size": whole[0], "sum_im_size": whole[1], "kept": whole[2],
                          "equals_sum_over_ranks": whole == [float(v) for v in totals[:3]]}
        if not sharding_check["equals_sum_over_ranks"]:
            raise SystemExit(f"sharded IM stage differs from the single-rank one: {whole} vs {totals[:3].tolist()}")

    # ---- roofline of the dominant kernel family: the one with the largest summed time over ALL hooked kernels ---------
    v = max(range(NV), key=lambda i: pms[i])
    fam_all = {FAMILIES[i]: {"launches": int(pc[i]), "ms": round(pms[i], 3), "avg_us": round(1000 * pms[i] / pc[i], 2),
                             "GBps": round(pby[i] / pms[i] / 1e6, 1) if pms[i] else None,
                             "share_of_sampled_time": round(pms[i] / max(sum(pms), 1e-9), 3)}
               for i in range(NV) if pc[i]}
    achieved = pby[v] / pms[v] / 1e6 if pms[v] else 0.0      # bytes / ms / 1e6 = GB/s
    # HBM traffic per launch from the PMC passes of profiles/collect.sh (FETCH_SIZE x 2 + WRITE_SIZE, separate passes)
    traffic, traffic_src = None, None
    try:
        import csv
        fam = FAMILIES[v].split("<")[0].split("+")[0].replace("wgf_stage1", "wgf_stage")
        for name in ("r02_pmc_traffic.csv", "r01_pmc_traffic.csv"):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            rows = [r for r in csv.DictReader(open(path)) if r["kernel"].startswith(fam)]
            nl = sum(int(r["launches"]) for r in rows)
            if nl:
                traffic = round(1e6 * sum(int(r["launches"]) * float(r["hbm_MB_per_launch_corrected(2*fetch+write)"]) for r in rows) / nl)
                traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, launch-weighted mean)"
                break
    except Exception:
        pass
    whole_n = setup_prof[0][v] + pc[v]
    whole_ms = setup_prof[1][v] + pms[v]
    steps_per_gen = max(info.get("epoch_steps", 1), 1)
    # whole-stage view (SURVEY 8d minimum bytes / stage time): what the dependent launch chains cost beyond the kernels
    inf_bytes = U * (N_MODELS * FWD_MIN_BYTES_PER_IMAGE + IM_BYTES_PER_IMAGE)
    step_bytes = 3 * BATCH * FWD_MIN_BYTES_PER_IMAGE
    step_view = {"ensemble_infer_plus_im": {"min_bytes": inf_bytes, "ms": round(t_inf, 3),
                                            "GBps": round(inf_bytes / t_inf / 1e6, 1), "frac": round(inf_bytes / t_inf / 1e6 / HBM_PEAK_GBS, 4)},
                 "train_step": {"min_bytes": step_bytes, "ms": round(t_ep / steps_per_gen, 4),
                                "GBps": round(step_bytes / (t_ep / steps_per_gen) / 1e6, 1),
                                "frac": round(step_bytes / (t_ep / steps_per_gen) / 1e6 / HBM_PEAK_GBS, 4)},
                 "note": "SURVEY 8d minimum HBM bytes (10 551 296 B per image and model forward, 3x per training image, "
                         "1 MiB per image for the IM chain) over the measured stage time, per rank"}
    roofline = {"bound": "hbm", "kernel": FAMILIES[v], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_us_per_launch_whole_process": round(1000 * whole_ms / max(whole_n, 1), 2),
                "launches": int(pc[v]), "avg_us_per_launch": round(1000 * pms[v] / max(pc[v], 1), 2),
                "avg_algorithmic_bytes_per_launch": round(pby[v] / max(pc[v], 1)),
                "share_of_sampled_kernel_time": round(pms[v] / max(sum(pms), 1e-9), 3),
                "sampling": f"every {args.prof_period}th hooked kernel launch of the timed region (all families)",
                "all_families": fam_all, "step": step_view,
                "note": "timed region: the weight-gradient kernels (side stream) and the ensemble's second model run beside "
                        "the main-stream kernels, event-bracketed durations include that sharing; 'exclusive' = the same "
                        "workload with every kernel alone on one stream (imk_debug_single_stream)"}
    # the same generation once more with every kernel alone on the stream: the kernels' own rates
    imk_lib.imk_debug_single_stream(1)
    imk_lib.imk_prof_enable(7 if not args.no_prof else 0)     # outside the timed region: dense sampling
    prof_collect()
    generation()
    barrier()
    xc, xms, xby = prof_collect()
    imk_lib.imk_prof_enable(0)
    imk_lib.imk_debug_single_stream(0)
    if xms[v]:
        x_ach = xby[v] / xms[v] / 1e6
        roofline["exclusive"] = {"achieved": round(x_ach, 1), "frac": round(x_ach / HBM_PEAK_GBS, 4),
                                 "launches": int(xc[v]), "avg_us_per_launch": round(1000 * xms[v] / max(xc[v], 1), 2),
                                 "all_families_GBps": {FAMILIES[i]: round(xby[i] / xms[i] / 1e6, 1) for i in range(NV) if xms[i]}}

    # ---- the fused IM kernel on the same shapes (HBM-bound; SURVEY 8d: 1 MiB / image) ------------------------
    nb = min(args.infer_batch, U)
    probs = torch.stack([m.predict_device(x_unl[:nb]) for m in models], 0)
    for _ in range(3):
        imk_im.im_binary(probs, 0.5, False, x_unl[:nb], True, True)
    a, b = ev(), ev()
    n_rep = 20
    a.record()
    for _ in range(n_rep):
        imk_im.im_binary(probs, 0.5, False, x_unl[:nb], True, True)
    b.record()
    torch.cuda.synchronize()
    im_ms = a.elapsed_time(b) / n_rep
    im_bytes = probs.shape[1] * (N_MODELS * H * W * K * 4 + 2 * H * W * C + 2 * H * W)
    im_kernel = {"kernel": "im_binary_vec<1>", "GBps": round(im_bytes / im_ms / 1e6, 1),
                 "frac_of_hbm_peak": round(im_bytes / im_ms / 1e6 / HBM_PEAK_GBS, 4),
                 "bytes_per_launch": im_bytes, "ms_per_launch": round(im_ms, 4)}

    if rank == 0:
        n_images = U_total if strong else U_total * world
        out = {
            "metric": "images/sec per IM generation (ensemble infer + IM build + 1 train epoch), 256x256",
            "value": round(n_images * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "ISIC-2018 binary 256x256x3, 2-model IM ensemble, tiny U-Net alpha=0.5 on MI355X "
                                   "(configs[1])", "unlabeled_images": n_images, "unlabeled_images_per_gpu": U,
                       "labeled_images": U_LABELED if strong else U_LABELED * world,
                       "n_models": N_MODELS, "infer_batch": args.infer_batch, "train_batch_per_gpu": BATCH,
                       "global_batch": BATCH * world, "parallelism": f"dp{world}",
                       **{k: v for k, v in info.items() if k != "n_train"}},
            "stage_ms": {"ensemble_infer_plus_im": round(t_inf, 2), "train_epoch": round(t_ep, 2)},
            "roofline": roofline,
            "im_kernel": im_kernel,
        }
        if sharding_check:
            out["sharding_check"] = sharding_check
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is a 1-GPU-run item (rank 0 only)
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["parity_sample"] = parity_sample(models, x_unl)
            out["png_io"] = png_io_rate(x_unl[:64].cpu().numpy())
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
