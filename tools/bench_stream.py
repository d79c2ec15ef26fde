#!/usr/bin/env python3
"""Streaming throughput (BASELINE.json configs[3]-style: many concurrent real-time streams, frame by frame) on one MI355X.

    python tools/bench_stream.py [--streams 4096] [--frames-per-call 1 4 16] [--calls 200]

Prints one JSON line per frames-per-call setting: hops/s over all streams, ms per call, and the number of real-time 48 kHz streams
one GPU sustains at that call size (a stream needs 100 hops/s)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--frames-per-call", type=int, nargs="+", default=[1, 4, 16])
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--model", default="df3", choices=["df3", "df3_ll", "defaults"],
                    help="df3_ll: DeepFilterNet3 without lookahead (the reference's low-latency LADSPA model, ladspa/README.md:3)")
    ap.add_argument("--gating", action="store_true", help="per-stream stage gating + silent-input shortcut (tract.rs:513-525,658-672)")
    args = ap.parse_args()
    from deepfilternet_amd import _lib
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.state_dict import random_state_dict
    from deepfilternet_amd.streaming import DfStream

    p = ModelParams.defaults() if args.model == "defaults" else ModelParams.deepfilternet3()
    if args.model == "df3_ll":
        p.conv_lookahead = p.df_lookahead = 0
    model, df_state, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
    dev = _lib.device()
    for n in args.frames_per_call:
        rt = DfStream(model, df_state, streams=args.streams, max_frames=n, gating=args.gating)
        hop = rt.frame_length
        x = 0.1 * torch.randn((args.streams, n * hop), device=dev)
        for _ in range(10):
            rt.process(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev_sync = int(os.environ.get("DFX_BENCH_DEV_SYNC", "0"))   # dev: the host waits after every n-th call (enqueue depth experiment)
        for i in range(args.calls):
            y = rt.process(x)
            if dev_sync and (i + 1) % dev_sync == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(y).all()
        # host cost of a call: 32 calls enqueued into empty queues, no wait in between (what bounds the rate on a box with a slow host)
        t1 = time.perf_counter()
        for i in range(32):
            y = rt.process(x)
        host_ms = (time.perf_counter() - t1) / 32 * 1e3
        torch.cuda.synchronize()
        hops = args.streams * n * args.calls
        ms_call = dt / args.calls * 1e3
        print(json.dumps({"metric": "streaming 48 kHz hops/s over all streams", "value": hops / dt, "unit": "frames/s", "streams": args.streams,
                          "frames_per_call": n, "ms_per_call": ms_call, "host_ms_per_call": host_ms, "call_budget_ms": 10.0 * n,
                          "realtime_streams_per_gpu": int(hops / dt / 100.0), "model": args.model, "gating": bool(args.gating),
                          "algorithmic_latency_ms": (p.fft_size - p.hop_size + rt.delay_frames * p.hop_size) / p.sr * 1e3}), flush=True)
        del rt


if __name__ == "__main__":
    main()
