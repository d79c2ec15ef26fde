"""Dev: is the STFT kernel alone enough to reproduce the two-handle corruption?  Two host threads, a DF state and a torch stream each, call
dfx_features (dfx_k_analysis + dfx_k_norm_scan4) in a loop on their own inputs / outputs and compare every result with their solo result.

  python tools/dev/two_analysis.py [--B 256] [--T 96960] [--iters 200] [--other analysis|matmul|copy|none] [--streams own|shared]
"""
import argparse
import os
import sys
import threading

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--T", type=int, default=96960)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--other", default="analysis")
ap.add_argument("--streams", default="own")
ap.add_argument("--victim", default="analysis")
args = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from deepfilternet_amd.enhance import df_features  # noqa: E402
from deepfilternet_amd.libdf import DF  # noqa: E402

B, T = args.B, args.T
states = [DF(48000, 960, 480, 32, 2) for _ in range(2)]
xs = [torch.from_numpy((0.1 * np.random.default_rng(1 + i).standard_normal((B, T))).astype(np.float32)).cuda() for i in range(2)]
refs = []
for i in range(2):
    s, e, f = df_features(xs[i], states[i], 96)
    torch.cuda.synchronize()
    refs.append((s.clone(), e.clone(), f.clone()))
streams = [torch.cuda.Stream() for _ in range(2)] if args.streams == "own" else [torch.cuda.current_stream()] * 2
bar = threading.Barrier(2)
bad = [[], []]
stop = threading.Event()


def describe(i, it, now, ref, name):
    a, b = now.flatten().view(torch.int32), ref.flatten().view(torch.int32)
    ne = (a != b).nonzero().flatten()
    if ne.numel() == 0:
        return
    row = now.shape[-2] * now.shape[-1] if now.dim() >= 5 else now.shape[-1]
    idx = ne.tolist()
    runs, start, prev = [], idx[0], idx[0]
    for j in idx[1:]:
        if j != prev + 1:
            runs.append((start, prev))
            start = j
        prev = j
    runs.append((start, prev))
    Tf = now.shape[2] if now.dim() >= 4 else now.shape[1]
    txt = []
    for (a0, a1) in runs[:6]:
        r, c0 = divmod(a0, row)
        txt.append(f"clip {r // Tf} frame {r % Tf} floats {c0}..{c0 + a1 - a0} (byte addr mod 128 = {(now.data_ptr() + 4 * a0) % 128})")
    print(f"  thread {i} iter {it} {name}: {ne.numel()} floats in {len(runs)} runs: " + "; ".join(txt), flush=True)
    a0 = runs[0][0]
    print(f"    now {now.flatten()[a0:a0 + 6].tolist()}\n    ref {ref.flatten()[a0:a0 + 6].tolist()}", flush=True)
    # per frame: which workgroup / wave of dfx_k_analysis computed it (grid-stride: 8 frames per workgroup pass, grid = min(frames / 8, 9 * CUs))
    nfr = now.shape[0] * Tf
    grid = min((nfr + 7) // 8, 9 * torch.cuda.get_device_properties(0).multi_processor_count)
    fr = torch.unique(ne // row).tolist()
    per = {}
    for f in fr:
        seg = ne[(ne // row) == f] - f * row
        bins = torch.unique(seg // 2).tolist()
        comp = sorted(set((seg % 2).tolist()))
        per.setdefault(((f // 8) % grid), []).append((f, f % 8, len(bins), bins[0], bins[-1], comp))
    for wg, lst in list(per.items())[:6]:
        print(f"    workgroup {wg}: " + "; ".join(f"frame {f} (clip {f // Tf} t {f % Tf}) wave {t}: {n} bins {b0}..{b1} parts {c}" for f, t, n, b0, b1, c in lst[:6]), flush=True)


def victim_run(i):
    if args.victim == "torchfft":
        return torch.view_as_real(torch.fft.rfft(xs[i].view(B * (T // 960), 960), dim=-1))
    if args.victim == "torchelem":
        v = xs[i]
        for _ in range(6):
            v = torch.sin(v * 1.0001 + 0.5) * torch.cos(v)
        return v
    return df_features(xs[i], states[i], 96)[0]


vref = None


def analysis_loop(i):
    global vref
    with torch.cuda.stream(streams[i]):
        if args.victim != "analysis":
            vref = victim_run(i).clone()
            streams[i].synchronize()
        bar.wait()
        for it in range(args.iters):
            if args.victim != "analysis":
                ok = torch.equal(victim_run(i), vref)
                if not ok:
                    bad[i].append(it)
                continue
            s, e, f = df_features(xs[i], states[i], 96)
            ok = torch.equal(s, refs[i][0])
            if not ok:
                bad[i].append(it)
                if len(bad[i]) <= 3:
                    describe(i, it, s, refs[i][0], "spec")
    stop.set()


def other_loop(i):
    if args.other in ("enhance", "forward"):
        from deepfilternet_amd.config import ModelParams
        from deepfilternet_amd.enhance import enhance, init_df
        from deepfilternet_amd.state_dict import random_state_dict
        pp = ModelParams.deepfilternet3()
        model, st = init_df(params=pp, state_dict=random_state_dict(pp, 0), epoch="none")[:2]
        n = 0
        with torch.cuda.stream(streams[i]):
            if args.other == "forward":
                sp, fe, fs = df_features(xs[i], st, 96)
            bar.wait()
            while not stop.is_set():
                if args.other == "forward":
                    model(sp, fe, fs)
                else:
                    enhance(model, st, xs[i])
                n += 1
                if n % 4 == 0:
                    streams[i].synchronize()
        try:
            model.check()
        except Exception as e:   # noqa: BLE001
            print("other handle:", repr(e)[:200])
        print(f"other loop: {n} passes, persistent {model.query(1)}")
        return
    with torch.cuda.stream(streams[i]):
        a = torch.randn(4096, 4096, device="cuda")
        big = torch.empty(64 << 20, device="cuda")
        bar.wait()
        while not stop.is_set():
            if args.other == "matmul":
                (a @ a).sum().item()
            elif args.other == "bf16mm":
                ab = a.to(torch.bfloat16)
                for _ in range(50):
                    ab @ ab
                streams[i].synchronize()
            elif args.other == "streamcopy":
                for _ in range(20):
                    big.copy_(big.flip(0))
                streams[i].synchronize()
            elif args.other == "copy":
                big.copy_(big.flip(0))
                streams[i].synchronize()
            else:
                import time
                time.sleep(0.01)


ts = [threading.Thread(target=analysis_loop, args=(0,)),
      threading.Thread(target=analysis_loop if args.other == "analysis" else other_loop, args=(1,))]
[t.start() for t in ts]
[t.join() for t in ts]
torch.cuda.synchronize()
print(f"SUMMARY other={args.other} streams={args.streams}: wrong iterations {[len(b) for b in bad]} of {args.iters} (first: {[b[:5] for b in bad]})")
