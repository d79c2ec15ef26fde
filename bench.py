#!/usr/bin/env python3
"""Benchmark of the enhance() hot path on MI355X (BASELINE.json metric: 48 kHz audio frames/s, hop = 480).

    python bench.py --gpus N --steps K --warmup W

One "step" = one enhance() over one batch of B synthetic noisy 48 kHz clips already resident in HBM
(config.workload: DeepFilterNet3, B=256 x 10 s per GPU = BASELINE.json configs[1]; weak scaling over GPUs: every rank
owns its own B clips, no data-path collective; the finished waveforms are gathered to rank 0 over RCCL, overlapped with
the next step).  Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      the north-star DF-apply kernel (fused deep filter + ERB gains): algorithmic bytes / hipEvent-timed launch
  kernels       hipEvent-timed per-kernel breakdown of one extra (untimed) step with the stream-level concurrency off
  cpu_baseline  the CPU oracle (oracle/, a port of the reference path) timed on this host on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # one hardware queue per engine stream (read when HIP initialises)
REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SR, HOP, FFT = 48000, 480, 960


def synth_audio(B: int, T: int, seed: int, device) -> torch.Tensor:
    """SURVEY.md §8(d) recipe: 5 harmonics of a 100-300 Hz f0 with a 4 Hz AM envelope (amplitude 0.1) + white noise at
    0 dB SNR, clipped to [-1, 1]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    f0 = 100.0 + 200.0 * torch.rand((B, 1), generator=g)
    ph = 2 * np.pi * torch.rand((B, 5), generator=g)
    f0, ph = f0.to(device), ph.to(device)
    t = torch.arange(T, device=device, dtype=torch.float32)[None, :] / SR
    x = torch.zeros((B, T), device=device)
    for h in range(5):
        x += torch.sin(2 * np.pi * f0 * (h + 1) * t + ph[:, h:h + 1]) / (h + 1)
    x *= 0.5 * (1 + torch.sin(2 * np.pi * 4.0 * t))
    x *= 0.1 / x.pow(2).mean(dim=1, keepdim=True).sqrt().clamp_min(1e-9)
    gd = torch.Generator(device=device).manual_seed(seed + 1)
    x += 0.1 * torch.randn((B, T), device=device, generator=gd)
    return x.clamp_(-1, 1).contiguous()


def cpu_baseline(p, sd, clips: int, seconds: float) -> dict:
    """The oracle's enhance() (C port of libDF, sequential over channels like pyDF, + torch-CPU DeepFilterNet3) on a bounded
    sample of the same workload."""
    from oracle import dfnet_oracle as O

    T = int(seconds * SR)
    x = synth_audio(clips, T, 1234, torch.device("cpu")).numpy()
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    O.enhance(p, sdt, x[:1, : SR // 2])  # warm-up (library load, thread pool)
    t0 = time.perf_counter()
    y = O.enhance(p, sdt, x)
    dt = time.perf_counter() - t0
    assert y.shape == x.shape
    frames = clips * (T // HOP)
    return {"value": frames / dt, "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{clips} clips x {seconds:g} s ({frames} frames, {dt:.2f} s wall): oracle/ C port of libDF "
                      f"(single thread, sequential over channels like pyDF) + torch-CPU DfNet3 ({torch.get_num_threads()} "
                      "threads); the reference's Rust libDF/tract cannot be built here (no cargo)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0, help="clip length")
    ap.add_argument("--model", default="df3", choices=["df3", "defaults"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the final RCCL gather of the waveforms to rank 0")
    ap.add_argument("--cpu-clips", type=int, default=32)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: PLW0642

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from deepfilternet_amd import _lib
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.state_dict import random_state_dict

    _lib.use_library(_lib.DEFAULT_LIB) if os.path.exists(_lib.DEFAULT_LIB) else _lib.lib()
    assert not _lib.is_emulator()
    p = ModelParams.deepfilternet3() if args.model == "df3" else ModelParams.defaults()
    sd = random_state_dict(p, 0)
    model, df_state, _, _ = init_df(params=p, state_dict=sd, epoch="none")
    B, T = args.batch, int(args.seconds * SR)
    Tf = (T + FFT) // HOP  # frames the kernels process per clip (pad=True)
    x = synth_audio(B, T, 100 + rank, dev)

    gather = world > 1 and not args.no_gather
    from deepfilternet_amd.distributed import enhance_sharded

    pending = [None, None]  # the gather of step i overlaps step i+1 (RCCL runs on its own stream)

    def step(i: int):
        k = i & 1
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        if gather:
            pending[k] = enhance_sharded(model, df_state, x, presharded=True, counts=[B] * world, gather=True, dst=0)
            return pending[k].local
        return enhance(model, df_state, x)

    def drain():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(["dfx_k_df_apply"])  # two hipEventRecords per step on the launch stream; everything else untouched
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        y = step(i)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dfa_ms, dfa_n = _lib.prof_read().get("dfx_k_df_apply", (0.0, 0))
    _lib.prof_enable(None)
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(y).all()
    model.check()  # raises if a workgroup pair of the two-CU GRU kernel ever timed out (results would be invalid)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the north-star kernel (fused deep filter + ERB gains over the full spectrum)
    F, E, O, nd = p.freq_bins, p.nb_erb, p.df_order, p.nb_df
    bytes_per_frame = F * 8 + nd * O * 8 + E * 4 + F * 8  # read X, read coefs, read gains, write Y   
    alg_bytes = bytes_per_frame * B * Tf
    roofline = None
    if dfa_n:
        avg_ms = dfa_ms / dfa_n
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "df_apply_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("batch") == B and tj.get("frames_per_clip") == Tf and tj.get("model") == args.model:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        roofline = {"kernel": "dfx_k_df_apply", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4), "launches": dfa_n}

    # ---- per-kernel breakdown of one extra, untimed step, branches serialised so that kernel times do not overlap
    _lib.prof_reset()
    _lib.prof_enable("all")
    model.set_streams(False)
    enhance(model, df_state, x)
    torch.cuda.synchronize()
    model.set_streams(True)
    kern = {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in sorted(_lib.prof_read().items(), key=lambda kv: -kv[1][0])}
    _lib.prof_enable(None)

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(p, sd, args.cpu_clips, args.seconds)
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)}

    frames = world * B * (T // HOP) * args.steps
    out = {
        "metric": "48 kHz audio frames/sec (hop=480), DeepFilterNet3 enhance()",
        "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded harmonic+noise 48 kHz audio, seeded random DeepFilterNet3 weights)",
        "config": {"workload": f"DeepFilterNet3 ({'recalled shipped shape, conv_ch=64' if args.model == 'df3' else 'code defaults'}) "
                               f"enhance(pad=True), batch={B} clips x {args.seconds:g} s @48 kHz per GPU, {Tf} STFT frames per clip",
                   "batch_per_gpu": B, "clip_seconds": args.seconds, "global_batch": B * world,
                   "parallelism": f"clips sharded over {world} GPU(s); " + ("async RCCL gather of waveforms to rank 0" if gather else "no collective"),
                   "inputs_resident_in_hbm": True},
        "roofline": roofline, "cpu_baseline": cpu, "kernels": kern,
        "realtime_factor": frames / dt / 100.0,
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
