#!/usr/bin/env python3
"""dev: what the PCIe path of this box does — page-locked H2D / D2H alone, both at once, and under a running enhance() loop."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

B, T = 256, 480000
n = B * T * 4
dev = torch.device("cuda", 0)
xh = torch.empty((B, T), dtype=torch.float32, pin_memory=True).normal_()
yh = torch.empty((B, T), dtype=torch.float32, pin_memory=True)
xd = torch.empty((B, T), device=dev)
yd = torch.randn((B, T), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
out = {}

def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def h2d():
    with torch.cuda.stream(s1):
        xd.copy_(xh, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2):
        yh.copy_(yd, non_blocking=True)
def both():
    h2d(); d2h()
out["h2d_gbs"] = n / timeit(h2d) / 1e9
out["d2h_gbs"] = n / timeit(d2h) / 1e9
out["both_gbs_each"] = n / timeit(both) / 1e9
for sdma in ("default",):
    pass
if "--compute" in sys.argv:
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import enhance, init_df
    p = ModelParams.deepfilternet3()
    model, st, _, _ = init_df(params=p, epoch="none", seed=0)
    x = 0.1 * torch.randn((B, T), device=dev)
    def comp():
        enhance(model, st, x)
    out["compute_ms"] = timeit(comp, 5) * 1e3
    def comp_copy():
        both(); enhance(model, st, x)
    out["compute_with_copies_ms"] = timeit(comp_copy, 5) * 1e3
    def comp_h2d():
        h2d(); enhance(model, st, x)
    out["compute_with_h2d_ms"] = timeit(comp_h2d, 5) * 1e3
    def comp_d2h():
        d2h(); enhance(model, st, x)
    out["compute_with_d2h_ms"] = timeit(comp_d2h, 5) * 1e3
print(json.dumps(out))
if "--variants" in sys.argv:
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import enhance, init_df
    import bench
    p = ModelParams.deepfilternet3()
    model, st, _, _ = init_df(params=p, epoch="none", seed=0)
    x = 0.1 * torch.randn((B, T), device=dev)
    main = torch.cuda.current_stream()
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    res = {}
    def v1():
        both(); enhance(model, st, x)
    res["v1_probe"] = timeit(v1) * 1e3
    def v2():
        both(); e1.record(s1); e2.record(s2); enhance(model, st, x)
    res["v2_events_recorded"] = timeit(v2) * 1e3
    def v3():
        both(); e1.record(s1); e2.record(s2); main.wait_event(e1); enhance(model, st, x)
    res["v3_compute_waits_for_upload"] = timeit(v3) * 1e3
    last = [enhance(model, st, x)]
    def v4():
        with torch.cuda.stream(s1):
            xd.copy_(xh, non_blocking=True)
        with torch.cuda.stream(s2):
            yh.copy_(last[0], non_blocking=True)
        last[0] = enhance(model, st, xd)
    torch.cuda.synchronize()
    res["v4_real_buffers_no_sync"] = timeit(v4) * 1e3
    def v5():
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            xd.copy_(xh, non_blocking=True)
        with torch.cuda.stream(s2):
            yh.copy_(last[0], non_blocking=True)
        last[0] = enhance(model, st, x)
    res["v5_host_sync_then_copies_then_pass"] = timeit(v5) * 1e3
    res["v6_bench_host_io"] = bench.bench_host_io(model, st, x, 6)["ms_per_step_host_to_host"]
    print(json.dumps(res))
