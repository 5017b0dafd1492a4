"""GPU A/B: per-tile conv kernel vs the persistent pipelined one (IMK_CONV_PIPE=0/1, one process each)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child():
    import torch, hashlib
    from inconsistencymasks_amd.prof import Profiler
    pr = Profiler(0)
    from inconsistencymasks_amd.unet import UNet
    torch.manual_seed(0)
    dev = "cuda"
    x = torch.randint(0, 256, (128, 256, 256, 3), dtype=torch.uint8, device=dev)
    y = (torch.rand((32, 256, 256, 1), device=dev) > 0.7).to(torch.uint8)
    m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=3)
    def prof(fn, n):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        pr.set_period(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        pr.set_period(0)
        c, ms, by, _ = pr.collect()
        tot = e0.elapsed_time(e1) / n
        return tot, {i: (int(c[i]) // n, round(ms[i] / n, 3), round(by[i] / ms[i] / 1e6) if ms[i] else 0) for i in range(7) if c[i]}
    p = m.predict_device(x)
    print("probs sha", hashlib.sha1(p.cpu().numpy().tobytes()).hexdigest()[:12])
    print("inference B=128: ms/call, {variant: (launches, ms, GB/s)}", prof(lambda: m.predict_device(x), 10))
    m.init_train_state()
    xs = x[:32].contiguous()
    m.fwd_bwd(xs, y, 0); torch.cuda.synchronize()
    print("grads sha", hashlib.sha1(m.grads.cpu().numpy().tobytes()).hexdigest()[:12], "stats", m.stats.cpu().tolist())
    print("train step B=32: ms/call", prof(lambda: m.train_step(xs, y, 0, 3e-3, 1e-4), 20))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for v in ("0", "1"):
            print("=== IMK_CONV_PIPE=" + v, flush=True)
            subprocess.run([sys.executable, __file__, "child"], env={**os.environ, "IMK_CONV_PIPE": v})
