#!/usr/bin/env python3
"""Benchmark of the enhance() hot path on MI355X (BASELINE.json metric: 48 kHz audio frames/s, hop = 480).

    python bench.py --gpus N --steps K --warmup W

One "step" = one enhance() over one batch of B synthetic noisy 48 kHz clips already resident in HBM
(config.workload: DeepFilterNet3, B=256 x 10 s per GPU = BASELINE.json configs[1]; weak scaling over GPUs: every rank
owns its own B clips, no data-path collective; the finished waveforms are gathered to rank 0 over RCCL, overlapped with
the next step).  Prints ONE JSON line on rank 0 (contract in the task statement), including
  dtype         "f32" data and accumulation; the dense contractions of the GRU projections / recurrences and of the fused DF-encoder
                convolutions run as fp16-split MFMAs (x = hi + lo in f16, 3 products, fp32 accumulate: ~2^-21 relative).
                exact_fp32_ms_per_step: the same step with every contraction on the exact fp32 kernels (DFX_EXACT_FP32=1), same run.
  roofline      the kernel of the TIMED LOOP that applies the deep filter: dfx_k_synthesis_rows (deep filter + ERB gains + ISTFT), algorithmic
                bytes / hipEvent-timed launch inside the timed loop; roofline.standalone_df_apply: the API kernel dfx_k_df_apply_rows timed
                alone at this size (enhance() does not launch it); rooflines: the other kernels SURVEY.md §8(d) prices (GRU recurrence alone /
                under load vs the matrix peaks, STFT vs HBM); summary (last key of the line): the headline numbers once more
  configs       driver-timed BASELINE.json configs[3] (4096 streams frame by frame, DeepFilterNet3 without lookahead) and configs[4]
                (deep filter of order 10 at batch 256: kernel roofline)
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy; 6.0 plain / 6.7 non-temporal here)
FP32_MATRIX_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 MFMA / VALU peak
F16_MFMA_PEAK_TF = 2500.0     # dense f16 MFMA peak (what the fp16-split kernels issue at: 3 MFMA flops per fp32-equivalent flop)
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


def macs_per_frame(p) -> dict:
    """Multiply-accumulates of ONE frame through the reference's arithmetic (what `DfNet.forward` + the deep filter compute, not what the
    fused kernels recompute), term by term (deepfilternet3.py:100-330, modules.py:18-126,702-780, multiframe.py:126-180)."""
    C, E, Fd, O, H = p.conv_ch, p.nb_erb, p.nb_df, p.df_order, p.emb_hidden_dim
    emb = C * E // 4
    g, ge = p.lin_groups, p.enc_lin_groups
    sep = lambda fout, k=3: fout * C * k + fout * C * C           # depthwise taps + pointwise C x C at fout positions
    gru = lambda layers: layers * 2 * H * 3 * H                    # W_ih + W_hh per layer
    m = {
        "erb_conv0": 9 * C * E, "erb_conv1": sep(E // 2), "erb_conv2": sep(E // 4), "erb_conv3": sep(E // 4),
        "df_conv0": 9 * 2 * (C // 2) * Fd // 1 + C * C * Fd,       # grouped 3x3 (2 -> C, groups 2) + pointwise
        "df_conv1": sep(Fd // 2),
        "df_fc_emb": (C * Fd // 2) * emb // ge, "enc_linear_in": emb * H // g, "enc_gru": gru(1), "enc_linear_out": H * emb // g, "lsnr": emb,
        "dec_linear_in": emb * H // g, "dec_gru": gru(p.emb_num_layers - 1), "dec_linear_out": H * emb // g,
        "convt3": sep(E // 4), "convt2": sep(E // 2), "convt1": sep(E), "conv0_out": 3 * C * E, "pathways": C * (E // 4 + E // 4 + E // 2 + E),
        "df_linear_in": emb * H // 8, "df_gru": gru(p.df_num_layers), "df_skip": (emb * H // g) if p.df_gru_skip == "groupedlinear" else 0,
        "df_out": H * (2 * O * Fd) // g,
        "df_convp": (C // 2) * p.df_pathway_kernel_size_t * 2 * O * Fd + (2 * O) * (2 * O) * Fd,   # groups = gcd(C, 2O) = 2, then 1x1
        "deep_filter": 4 * O * Fd + 2 * (p.fft_size // 2 + 1 - Fd),                                  # complex MACs as 4 real ones, gains as 2
    }
    m["total"] = sum(m.values())
    return m


def cpu_baseline(p, sd, clips: int, seconds: float, keep=None) -> dict:
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
    if keep is not None:   # the sample and the oracle's samples: the line's `parity` compares the engine with them
        keep["x"], keep["y"] = x, y
    frames = clips * (T // HOP)
    return {"value": frames / dt, "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{clips} clips x {seconds:g} s ({frames} frames, {dt:.2f} s wall): oracle/ C port of libDF "
                      f"(single thread, sequential over channels like pyDF) + torch-CPU DfNet3 ({torch.get_num_threads()} "
                      "threads); the reference's Rust libDF/tract cannot be built here (no cargo)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0, help="clip length")
    ap.add_argument("--model", default="df3", choices=["df3", "defaults"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the final RCCL gather of the waveforms to rank 0")
    ap.add_argument("--cpu-clips", type=int, default=32)
    ap.add_argument("--main-only", action="store_true", help="only the timed loop and the DF-apply roofline (profiling runs: no extra steps / configs)")
    ap.add_argument("--host-io-only", action="store_true", help="only the host-to-host pipeline (bench_host_io), one JSON line; the full run starts this in a process of its own")
    args = ap.parse_args()

    # ---- ranks: one process per GPU.  Under an external launcher (torch.distributed.run) WORLD_SIZE must equal --gpus; without one,
    # --gpus N > 1 starts the N ranks here (deepfilternet_amd.distributed.launch_ranks) and this process only waits for them.
    from deepfilternet_amd.distributed import WorldError, check_world, init_world, launch_ranks

    try:
        env_world = check_world(args.gpus, torch.cuda.device_count() if torch.cuda.is_available() else None)
    except WorldError as e:
        raise SystemExit(f"bench.py: {e}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the HIP engine has no CPU fallback")
    if env_world is None and args.gpus > 1:
        raise SystemExit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    world, rank, local_rank = env_world if env_world is not None else (1, 0, 0)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: PLW0642

        init_world("nccl", world, rank, dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        # one process per GPU, checked on what the ranks really hold (a launcher that did not apply LOCAL_RANK puts every rank on device 0)
        from deepfilternet_amd.distributed import check_distinct_devices, gather_device_ids

        try:
            check_distinct_devices(gather_device_ids(dev))
        except WorldError as e:
            raise SystemExit(f"bench.py: {e}")

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

    if args.host_io_only:
        print(json.dumps(bench_host_io(model, df_state, x, args.steps)), flush=True)
        return
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    _lib.prof_reset()
    # two hipEventRecords per step on the launch stream around the finishing kernel; everything else untouched.  (Since round 3 enhance()
    # applies the deep filter + gains INSIDE the ISTFT kernel, dfx_k_synthesis_rows; with DFX_FUSE_DFA=0 the separate deep-filter kernel is
    # timed in the loop as before.)
    # (round 6: and two around the STFT kernel, the first kernel of a pass: the serialised-step figure of `rooflines.dfx_k_analysis` runs it behind a
    # drained chip and is 0.03-0.08 ms slower than what the timed loop sees; A/B with and without these records, three runs each on one box: 12.514 / 12.099 / 12.100 vs
    # 12.401 / 12.081 / 12.067 ms per step, i.e. +0.02-0.03 ms inside the timed region — DFX_BENCH_PROF_ANALYSIS=0 leaves them out)
    loop_kernels = ["dfx_k_df_apply", "dfx_k_synthesis", "dfx_k_analysis"]
    if os.environ.get("DFX_BENCH_PROF_ANALYSIS") == "0":   # dev: the A/B of the comment above
        loop_kernels.remove("dfx_k_analysis")
    _lib.prof_enable(loop_kernels)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_sync = os.environ.get("DFX_BENCH_DEV_SYNC") == "1"   # dev diagnosis only (the host waits for every step: no enqueue-ahead)
    for i in range(args.steps):
        y = step(i)
        if dev_sync:
            torch.cuda.synchronize()
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loop_prof = _lib.prof_read()
    dfa_ms, dfa_n = loop_prof.get("dfx_k_df_apply", (0.0, 0))
    syn_ms, syn_n = loop_prof.get("dfx_k_synthesis", (0.0, 0))
    ana_ms, ana_n = loop_prof.get("dfx_k_analysis", (0.0, 0))
    _lib.prof_enable(None)
    per_rank_ms = [dt / args.steps * 1e3]
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in allt]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(y).all()
    gru_persistent = bool(model.query(model.Q_GRU_PERSISTENT))
    model.check()  # raises if a kernel of the loop reported a fault (fp16-split range, a flag wait that timed out): results would be invalid

    # ---- multi-GPU: the same timed loop without the final gather (so that a scaling run can tell compute from the collective)
    no_gather_ms = None
    if gather:
        for i in range(2):
            enhance(model, df_state, x)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            enhance(model, df_state, x)
        torch.cuda.synchronize()
        dist.barrier()
        dt2 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(dt2, op=dist.ReduceOp.MAX)
        no_gather_ms = float(dt2.item()) / args.steps * 1e3
    # ---- multi-GPU, the consumer is the host: no collective at all, every rank copies its own slice to page-locked host memory on
    # its stream (overlapping the next step like the gather does)
    host_consumer_ms = None
    if world > 1:
        hp = [None, None]
        for i in range(2 + args.steps):
            if i == 2:
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            k = i & 1
            if hp[k] is not None:
                hp[k].wait()
            hp[k] = enhance_sharded(model, df_state, x, presharded=True, counts=[B] * world, gather="host")
        for h in hp:
            if h is not None:
                h.wait()
        torch.cuda.synchronize()
        dist.barrier()
        dt3 = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
        dist.all_reduce(dt3, op=dist.ReduceOp.MAX)
        host_consumer_ms = float(dt3.item()) / args.steps * 1e3
        del hp

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the north-star kernel (fused deep filter + ERB gains over the full spectrum)
    F, E, O, nd = p.freq_bins, p.nb_erb, p.df_order, p.nb_df
    bytes_per_frame = F * 8 + nd * O * 8 + E * 4 + F * 8  # read X, read coefs, read gains, write Y   
    alg_bytes = bytes_per_frame * B * Tf

    if args.main_only:
        frames = world * B * (T // HOP) * args.steps
        print(json.dumps({"metric": "48 kHz audio frames/sec (hop=480), DeepFilterNet3 enhance()", "value": frames / dt, "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "dfa_in_loop_ms": (dfa_ms / dfa_n) if dfa_n else None, "finish_in_loop_ms": (syn_ms / syn_n) if syn_n else None,
                          "analysis_in_loop_ms": (ana_ms / ana_n) if ana_n else None,
                          "gru_phase_form": "persistent" if gru_persistent else "events", "exact_fp32": bool(model.query(model.Q_EXACT_FP32))}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-kernel breakdown of one extra, untimed step, branches serialised so that kernel times do not overlap
    _lib.prof_reset()
    _lib.prof_enable("all")
    model.set_streams(False)
    enhance(model, df_state, x)
    torch.cuda.synchronize()
    model.set_streams(True)
    serial = _lib.prof_read()
    kern = {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in sorted(serial.items(), key=lambda kv: -kv[1][0])}
    _lib.prof_enable(None)

    if os.environ.get("DFX_BENCH_SKIP_EXTRAS") == "1":   # dev (tools/gpu_kern.sh): the timed loop and the per-kernel breakdown only
        frames = world * B * (T // HOP) * args.steps
        print(json.dumps({"value": frames / dt, "ms_per_step": dt / args.steps * 1e3, "dfa_in_loop_ms": (dfa_ms / dfa_n) if dfa_n else None,
                          "kernels": kern}), flush=True)
        return

    def hbm_record(kernel, ms, nbytes, extra=None):
        ach = nbytes / (ms * 1e-3) / 1e9
        r = {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
             "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": round(ms, 4)}
        r.update(extra or {})
        return r

    # ---- roofline of the north-star kernel, dfx_k_df_apply_rows (fused deep filter + ERB gains over the full spectrum: the kernel behind
    # dfx_df_apply / dfx_model_forward).  enhance() no longer launches it (the same arithmetic runs inside the ISTFT kernel, see
    # rooflines.dfx_k_synthesis_rows), so it is timed here on this workload's own buffers: stand-alone launches on the launch stream with
    # the library's hipEvents around each (dfx_prof_*), in this process, right after the timed loop.
    def pmc_traffic(fname, tool):
        """HBM bytes per launch from the newest committed PMC measurement of this kernel at this size (PMC passes need rocprofv3 around the
        process: tools/gpu_pmc_*.sh; separate --pmc passes for FETCH_SIZE and WRITE_SIZE, calibrated on pure-stream dispatches)."""
        for rnd in ("r06_", "r05_", "r04_", "r03_", ""):
            tpath = os.path.join(REPO, "profiles", rnd + fname)
            if not os.path.exists(tpath):
                continue
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("batch") == B and tj.get("frames_per_clip") == Tf and tj.get("model", args.model) == args.model:
                    return tj.get("hbm_bytes_per_launch"), (f"static: profiles/{rnd}{fname} — rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes ({tool}) over this kernel "
                                                            "at this size, calibrated on pure-stream dispatches of the same kernel; not measured in this run "
                                                            "(PMC needs rocprofv3 around the process)")
            except Exception:  # noqa: BLE001
                pass
        return None, None

    traffic, traffic_src = pmc_traffic("df_apply_traffic.json", "tools/gpu_pmc_dfa.sh")
    roofline = None
    try:
        sa_ms, sa_n = bench_df_apply_rows(dev, df_state, B, Tf, F, p.nb_df, O, p.df_lookahead, E)
        roofline = hbm_record("dfx_k_df_apply_rows (stand-alone)", sa_ms / sa_n, alg_bytes,
                              {"traffic": traffic, "traffic_source": traffic_src, "launches": sa_n,
                               "where": "stand-alone launches of dfx_df_apply_strided at this workload's size (engine layout: 488-bin rows, tap-major coefficients), "
                                        "library hipEvents on the launch stream, inside bench.py after the timed loop"})
    except Exception as e:  # noqa: BLE001
        roofline = {"error": repr(e)}
    cfg_o10 = None
    if world == 1:   # BASELINE.json configs[4], same kind of stand-alone launches (before the extras below create more streams / pinned buffers)
        try:
            cfg_o10 = bench_df_apply_o10(dev, df_state, B, Tf)
        except Exception as e:  # noqa: BLE001
            cfg_o10 = {"error": repr(e)}
    if dfa_n and isinstance(roofline, dict) and "error" not in roofline:   # DFX_FUSE_DFA=0: the kernel also runs inside the loop
        roofline["in_loop"] = {"avg_launch_ms": round(dfa_ms / dfa_n, 4), "frac": round(alg_bytes / (dfa_ms / dfa_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- the other kernels SURVEY.md §8(d) prices
    rooflines = {}
    nfr = B * Tf
    fused_finish = not dfa_n   # enhance() finished with dfx_k_synthesis_rows (deep filter + gains + ISTFT in one kernel)
    fin_bpf = (F * 8 + p.nb_df * O * 8 + E * 4 + 480 * 4) if fused_finish else (F * 8 + 480 * 4)
    fin_name = "dfx_k_synthesis_rows" if fused_finish else "dfx_k_synthesis"
    for name, key, bpf in (("dfx_k_analysis", "dfx_k_analysis", 480 * 4 + F * 8 + E * 4), (fin_name, "dfx_k_synthesis", fin_bpf)):
        if key in serial and serial[key][1]:
            rooflines[name] = hbm_record(name, serial[key][0] / serial[key][1], bpf * nfr, {"where": "serialised step"})
    if ana_n and "dfx_k_analysis" in rooflines:
        ana_bpf = 480 * 4 + F * 8 + E * 4
        rooflines["dfx_k_analysis"]["in_loop"] = {"avg_launch_ms": round(ana_ms / ana_n, 4), "launches": ana_n,
                                                  "frac": round(ana_bpf * nfr / (ana_ms / ana_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                  "where": "inside the timed loop (hipEvents on the launch stream, one launch per step)"}
    if syn_n and fin_name in rooflines:
        rooflines[fin_name]["in_loop"] = {"avg_launch_ms": round(syn_ms / syn_n, 4), "launches": syn_n,
                                          "frac": round(fin_bpf * nfr / (syn_ms / syn_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "where": "inside the timed loop (hipEvents on the launch stream, one launch per step)"}
    if fused_finish and fin_name in rooflines:
        ft, fts = pmc_traffic("finish_traffic.json", "tools/gpu_pmc_finish.sh")
        rooflines[fin_name]["traffic"], rooflines[fin_name]["traffic_source"] = ft, fts
        rooflines[fin_name]["algorithmic_bytes_per_frame"] = fin_bpf
        rooflines[fin_name]["note"] = ("read X 3848 + coefficients 3840 + gains 128, write 1920 bytes of audio per frame; the enhanced spectrum "
                                       "(7696 B per frame written + read back by the two-kernel form) never exists in HBM")
    nlayers = 1 + (p.emb_num_layers - 1) + p.df_num_layers
    pair_form = os.environ.get("DFX_GRU_PAIR", "1")[:1] != "0" and B > 16 and os.environ.get("DFX_EXACT_FP32", "0")[:1] != "1"
    flop_step = 2.0 * B * 256 * 768          # h[B,256] x W_hh^T[256,768] per time step and layer (fp32-equivalent flops)
    gru = {"kernel": "dfx_k_gru_rec_h3", "bound": "mfma", "unit": "TFLOP/s", "peak": FP32_MATRIX_PEAK_TF, "peak_f16_mfma": F16_MFMA_PEAK_TF,
           "flop_per_step_and_layer": flop_step, "layers": nlayers, "steps_per_layer": Tf,
           "note": "fp32-equivalent flops; the kernel issues 3 f16 MFMA flops per flop (fp16-split). One layer = B/16 workgroups (16 CUs at B=256): "
                   "a sequential chain, latency-bound by construction; 'frac' prices one layer-kernel against the whole chip's fp32 matrix peak. "
                   "'alone' is the one-CU-per-16-clips launch form of the serialised step (W_hh streamed from the L2); 'under_load' is the persistent "
                   "launch of the timed path: " + ("dfx_k_gru_seq_p2, pairs of CUs per 32 clips with W_hh resident and h exchanged every step (csrc/dfx_gru_pair.h)"
                                                   if pair_form else "dfx_k_gru_seq, one CU per 16 clips")}
    if "dfx_k_gru_rec" in serial and serial["dfx_k_gru_rec"][1]:
        us_alone = serial["dfx_k_gru_rec"][0] * 1e3 / (nlayers * Tf)
        gru["alone"] = {"us_per_step": round(us_alone, 3), "achieved": round(flop_step / us_alone / 1e6, 2),
                        "frac": round(flop_step / us_alone / 1e6 / FP32_MATRIX_PEAK_TF, 4), "where": "serialised step: one layer at a time"}
    # under load: three more steps with events around every recurrence launch (all layers and the background kernels in flight)
    _lib.prof_reset()
    _lib.prof_enable(["dfx_k_gru_rec"])
    for i in range(3):
        enhance(model, df_state, x)
    torch.cuda.synchronize()
    g_ms, g_n = _lib.prof_read().get("dfx_k_gru_rec", (0.0, 0))
    _lib.prof_enable(None)
    if g_n and g_n <= 3:
        # persistent form (dfx_k_gru_seq): ONE launch carries all layers for the whole sequence, flag waits included; the chain a
        # frame goes through is 3 layers deep, so the launch lasts T steps + the pipeline fill
        phase_ms = g_ms / g_n
        us_chain = phase_ms * 1e3 / Tf
        gru["under_load"] = {"launch_ms": round(phase_ms, 3), "us_per_frame_of_the_sequence": round(us_chain, 3),
                             "achieved": round(nlayers * flop_step * Tf / (phase_ms * 1e-3) / 1e12, 2),
                             "frac": round(nlayers * flop_step * Tf / (phase_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TF, 4), "launches": g_n,
                             "where": "3 extra steps of the normal pipeline: one persistent launch (all layers, " + str(nlayers * ((B + 15) // 16)) +
                                      " workgroups = CUs) with the projections / decoder tails on the rest of the chip; includes its flag waits"}
        gru["achieved"], gru["frac"] = gru["under_load"]["achieved"], gru["under_load"]["frac"]
    elif g_n:
        us_load = g_ms * 1e3 / (3 * nlayers * Tf)
        gru["under_load"] = {"us_per_step": round(us_load, 3), "achieved": round(flop_step / us_load / 1e6, 2),
                             "frac": round(flop_step / us_load / 1e6 / FP32_MATRIX_PEAK_TF, 4), "launches": g_n,
                             "where": "3 extra steps of the normal pipeline (layers concurrent, projections / decoder tails beside them)"}
        gru["achieved"], gru["frac"] = gru["under_load"]["achieved"], gru["under_load"]["frac"]
    rooflines["dfx_k_gru_rec_h3"] = gru
    # ---- the parsed `roofline` object = the kernel that applies the deep filter INSIDE the timed loop (dfx_k_synthesis_rows: deep filter +
    # ERB gains + ISTFT), timed live with hipEvents on the launch stream, one launch per timed step.  The stand-alone deep-filter kernel
    # (dfx_k_df_apply_rows, the kernel behind dfx_df_apply, which enhance() does not launch) is roofline.standalone_df_apply; its two headline
    # numbers are repeated as scalars so that a reader who keeps only the first level still sees them.
    standalone = roofline
    if syn_n and fin_name in rooflines:
        ach = fin_bpf * nfr / (syn_ms / syn_n * 1e-3) / 1e9
        roofline = {"kernel": fin_name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": rooflines[fin_name].get("traffic"),
                    "traffic_source": rooflines[fin_name].get("traffic_source"),
                    "where": "inside the timed loop (library hipEvents on the launch stream, one launch per timed step)",
                    "algorithmic_bytes_per_frame": fin_bpf, "algorithmic_bytes_per_launch": fin_bpf * nfr,
                    "avg_launch_ms": round(syn_ms / syn_n, 4), "launches": syn_n,
                    "note": "deep filter + ERB gains + ISTFT in one kernel: the kernel of the benchmarked path that does the north-star DF-apply arithmetic",
                    "standalone_df_apply_frac": standalone.get("frac") if isinstance(standalone, dict) else None,
                    "standalone_df_apply_ms": standalone.get("avg_launch_ms") if isinstance(standalone, dict) else None,
                    "standalone_df_apply": standalone}
    elif isinstance(roofline, dict) and "error" not in roofline:
        roofline["where"] = "DFX_FUSE_DFA=0: " + str(roofline.get("where"))
    if isinstance(roofline, dict) and "error" not in roofline:
        mm = macs_per_frame(p)
        step_flop = 2.0 * mm["total"] * nfr
        step_s = dt / args.steps
        roofline["step"] = {"flop": step_flop, "macs_per_frame": mm, "frames": nfr, "achieved_tflops": round(step_flop / step_s / 1e12, 2),
                            "frac_fp32_matrix": round(step_flop / step_s / 1e12 / FP32_MATRIX_PEAK_TF, 4),
                            "frac_f16_mfma": round(3.0 * step_flop / step_s / 1e12 / F16_MFMA_PEAK_TF, 4),
                            "note": "reference arithmetic of DfNet.forward + the deep filter (fp32-equivalent flops, STFT / ISTFT not counted) over the timed "
                                    "step; frac_f16_mfma prices the three f16 matrix flops the fp16-split kernels issue per flop"}

    # ---- exact fp32: the same step with every contraction on the exact fp32 MFMA / VALU kernels
    exact_ms, exact_diff = None, None
    extras = world == 1   # a scaling run (N > 1) keeps the other ranks waiting at the final barrier: the N = 1 line carries the extras
    try:
        if not extras:
            raise RuntimeError("N > 1: reported by the N = 1 run")
        os.environ["DFX_EXACT_FP32"] = "1"
        os.environ["DFX_QUIET"] = "1"   # (a second handle in this process shares hardware queues with the first: its probe may choose the event form — only its output is used)
        m_exact, _, _, _ = init_df(params=p, state_dict=sd, epoch="none")
        del os.environ["DFX_EXACT_FP32"]
        ye = enhance(m_exact, df_state, x)   # (the output only: this process holds two model handles by now, whose ~40 streams share hardware
        torch.cuda.synchronize()             #  queues — the exact step is timed in a process of its own below; the stream handshake of the
        del os.environ["DFX_QUIET"]          #  second handle runs at its first pass: quiet until then)
        exact_diff = float((ye - y).pow(2).mean().sqrt())
        del m_exact, ye
    except Exception as e:  # noqa: BLE001
        exact_diff = repr(e)
    finally:
        os.environ.pop("DFX_EXACT_FP32", None)
        os.environ.pop("DFX_QUIET", None)

    # ---- host to host: the reference's enhance() takes and returns CPU tensors (enhance.py:206-250).  Page-locked [B, T] input and output,
    # H2D of batch k+1 and D2H of batch k-1 on their own streams under the compute of batch k.
    host_io = None
    if extras:
        host_io = "pending"   # run below in a process of its own (this one has created three model handles and ~40 streams by then)

    # ---- the CPU baseline (the oracle on a bounded sample) and, on the same clips, the parity of the timed handle
    cpu, parity = None, None
    if world == 1 and not args.no_cpu_baseline:
        try:
            keep = {}
            cpu = cpu_baseline(p, sd, args.cpu_clips, args.seconds, keep)
            # parity in the run that is timed: the engine (the timed handle, the timed arithmetic) on the clips the oracle has just enhanced.
            # The bar of the north star is 1e-4 RMS on the waveform; the reference's own habit is a quality pin beside every run (df/scripts/test_df.py:67-77).
            yo = torch.from_numpy(keep["y"])
            yh = enhance(model, df_state, torch.from_numpy(keep["x"]).to(dev)).cpu()
            model.check()
            d = (yh - yo).double()
            row = d.pow(2).mean(dim=1).sqrt()
            parity = {"rms": float(d.pow(2).mean().sqrt()), "max_row_rms": float(row.max()), "max_abs": float(d.abs().max()),
                      "signal_rms": float(yo.double().pow(2).mean().sqrt()), "clips": int(yo.shape[0]), "bar": 1e-4,
                      "against": "oracle/ (C port of libDF + torch-fp32 DfNet3) on the cpu_baseline sample, the timed handle and arithmetic"}
            if not parity["rms"] < 1e-4:
                raise SystemExit(f"bench.py: parity {parity['rms']:.3e} RMS against the oracle (bar 1e-4): the timed numbers would be meaningless")
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            cpu = cpu if cpu is not None else {"error": repr(e)}
            parity = {"error": repr(e)}


    # ---- BASELINE.json configs[3] and configs[4] (the batch model's streams are released first: a process with more streams than
    # hardware queues makes them share queues, which serialises the streaming runtime's three branches)
    import gc

    del model
    gc.collect()
    torch.cuda.empty_cache()
    # ---- the same loop with free enqueue-ahead (DFX_ENQUEUE_AHEAD=1: the host queues pass k+1, and the GRU phase of pass k, as early as it
    # can): what the engine's one-pass-in-flight / staged-phase enqueue policy is worth.  A process of its own, like the streaming
    # configuration below: this one has created three model handles by now (~40 streams; their hardware queues would be shared).
    ahead_ms = None
    exact_form = None
    try:
        import subprocess

        if not extras:
            raise RuntimeError("N > 1: reported by the N = 1 run")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                 "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT")}
        env["DFX_EXACT_FP32"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--main-only", "--steps", "5", "--warmup", "2", "--batch", str(B), "--seconds",
                            str(args.seconds), "--model", args.model], env=env, capture_output=True, text=True, timeout=600)
        je = json.loads(r.stdout.strip().splitlines()[-1])
        exact_ms, exact_form = je["ms_per_step"], je.get("gru_phase_form")
    except Exception as e:  # noqa: BLE001
        exact_ms = repr(e)
    try:
        import subprocess

        if not extras:
            raise RuntimeError("N > 1: reported by the N = 1 run")

        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                 "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT")}
        env["DFX_ENQUEUE_AHEAD"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--main-only", "--steps", "10", "--warmup", "2", "--batch", str(B), "--seconds",
                            str(args.seconds), "--model", args.model], env=env, capture_output=True, text=True, timeout=600)
        ahead_ms = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception as e:  # noqa: BLE001
        ahead_ms = repr(e)
    if host_io == "pending":
        try:
            import subprocess

            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                     "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT")}
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-io-only", "--steps", str(max(args.steps, 60)), "--batch", str(B), "--seconds",
                                str(args.seconds), "--model", args.model], env=env, capture_output=True, text=True, timeout=600)
            host_io = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:  # noqa: BLE001
            host_io = {"error": repr(e)}
    configs = {}
    if extras:
        try:
            configs["streaming_4096"] = bench_streaming(dev)
        except Exception as e:  # noqa: BLE001
            configs["streaming_4096"] = {"error": repr(e)}
        configs["df_apply_o10"] = cfg_o10
    torch.cuda.empty_cache()

    frames = world * B * (T // HOP) * args.steps
    out = {
        "metric": "48 kHz audio frames/sec (hop=480), DeepFilterNet3 enhance()",
        "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (storage and accumulation; GRU projections / recurrences, the fused DF-encoder convolutions and the pointwise halves of the ERB separable convolutions as fp16-split MFMAs: "
                 "x = hi + lo in f16, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16, ~2^-21 relative; see exact_fp32_ms_per_step)",
        "exact_fp32_ms_per_step": exact_ms, "exact_fp32_rms_diff_of_output": exact_diff, "exact_fp32_gru_phase_form": exact_form,
        "data": "synthetic (seeded harmonic+noise 48 kHz audio, seeded random DeepFilterNet3 weights)",
        "config": {"workload": f"DeepFilterNet3 ({'recalled shipped shape, conv_ch=64' if args.model == 'df3' else 'code defaults'}) "
                               f"enhance(pad=True), batch={B} clips x {args.seconds:g} s @48 kHz per GPU, {Tf} STFT frames per clip",
                   "batch_per_gpu": B, "clip_seconds": args.seconds, "global_batch": B * world,
                   "parallelism": f"clips sharded over {world} GPU(s); " + ("async RCCL gather of waveforms to rank 0" if gather else "no collective"),
                   "inputs_resident_in_hbm": True},
        "ms_per_step_without_gather": no_gather_ms, "ms_per_step_host_consumer": host_consumer_ms, "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
        "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
        "enqueue": {"policy": "one big pass in flight per model handle (a call first waits, on the host, for the previous pass to drain) and the GRU phase "
                              "of a pass is enqueued once its encoder front has run: packets waiting at the head of the pass's ~13 hardware queues slow "
                              "the kernels that are running; every step of the timed loop still runs to completion inside the timed region",
                    "ms_per_step_with_free_enqueue_ahead": ahead_ms, "switch": "DFX_ENQUEUE_AHEAD=1"},
        "host_io": host_io,
        "gru_phase_form": (("persistent flag-synchronised launch (dfx_k_gru_seq_p2: recurrences on pairs of CUs)" if os.environ.get("DFX_GRU_PAIR", "1")[:1] != "0" and B > 16 else "persistent flag-synchronised launch (dfx_k_gru_seq)") if gru_persistent else "event-synchronised launches per (layer, time chunk)"),
        "roofline": roofline, "rooflines": rooflines, "configs": configs, "cpu_baseline": cpu, "parity": parity, "kernels": kern,
        "realtime_factor": frames / dt / 100.0,
    }
    # the numbers a reader of the line's tail looks for, once more at the very end
    def _g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    out["summary"] = {
        "ms_per_step": round(dt / args.steps * 1e3, 3), "frames_per_s_fp16_split": round(frames / dt, 1),
        "exact_fp32_ms_per_step": exact_ms, "frames_per_s_exact_fp32": (round(world * B * (T // HOP) / (exact_ms * 1e-3), 1) if exact_ms else None),
        "exact_fp32_gru_phase_form": exact_form, "parity_rms_vs_oracle": _g(parity, "rms"),
        "roofline_kernel": _g(roofline, "kernel"), "roofline_frac": _g(roofline, "frac"),
        "standalone_df_apply_frac": _g(roofline, "standalone_df_apply_frac"),
        "analysis_frac": _g(rooflines, "dfx_k_analysis", "frac"), "analysis_frac_in_loop": _g(rooflines, "dfx_k_analysis", "in_loop", "frac"),
        "gru_under_load_us_per_frame_of_the_sequence": _g(rooflines, "dfx_k_gru_rec_h3", "under_load", "us_per_frame_of_the_sequence"),
        "gru_alone_us_per_step": _g(rooflines, "dfx_k_gru_rec_h3", "alone", "us_per_step"),
        "host_io_pcm16_over_resident": _g(host_io, "pcm16_over_resident"), "host_io_f32_over_resident": _g(host_io, "f32_over_resident"),
        "streaming_ms_per_call": _g(configs, "streaming_4096", "ungated", "ms_per_call"),
        "streaming_gated_ms_per_call": _g(configs, "streaming_4096", "stage_gating", "ms_per_call"),
        "df_apply_o10_frac": _g(configs, "df_apply_o10", "frac"),
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_streaming(dev, streams: int = 4096, calls: int = 1000) -> dict:
    """BASELINE.json configs[3]: DeepFilterNet3 without lookahead (the reference's low-latency LADSPA model, ladspa/README.md:3), 4096
    concurrent streams advanced frame by frame (one hop of every stream per call = df_process_frame for all of them), without and
    with the reference runtime's per-stream stage decisions (DfTract::process).  Runs tools/bench_stream.py in a process of its own:
    the batch model of this process holds ~25 HIP streams, and a process with more streams than hardware queues makes them share
    queues — the streaming runtime's three branches then serialise (1.7 ms per call instead of 0.9)."""
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    out = {"workload": f"DeepFilterNet3 without lookahead, {streams} streams x 1 hop (480 samples) per call, {calls} calls, inputs resident in HBM; "
                       "own process (tools/bench_stream.py)"}
    for tag, gating in (("ungated", False), ("stage_gating", True)):
        cmd = [sys.executable, os.path.join(here, "tools", "bench_stream.py"), "--model", "df3_ll", "--streams", str(streams), "--frames-per-call", "1",
               "--calls", str(calls)] + (["--gating"] if gating else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            out[tag] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        j = json.loads(line[-1])
        out[tag] = {k: j[k] for k in ("value", "unit", "ms_per_call", "call_budget_ms", "realtime_streams_per_gpu", "algorithmic_latency_ms")}
    return out


def bench_df_apply_rows(dev, df_state, B: int, Tf: int, F: int, nd: int, O: int, la: int, E: int, iters: int = 20):
    """The north-star kernel alone at the workload's size, on the engine's own layout (rows of 488 bins = 64-byte aligned, tap-major
    coefficients), timed by the library's hipEvents on the launch stream (dfx_prof_*).  -> (total ms, launches)."""
    from deepfilternet_amd import _lib

    Fs = (F + 7) // 8 * 8
    g = torch.Generator(device=dev).manual_seed(0)
    spec = torch.randn((B, Tf, Fs, 2), device=dev, generator=g)
    coefs = 0.3 * torch.randn((B, O, Tf, nd, 2), device=dev, generator=g)
    gains = torch.rand((B, Tf, E), device=dev, generator=g)
    out = torch.empty_like(spec)
    L = _lib.lib()

    def run():
        _lib.check(L.dfx_df_apply_strided(_lib.ptr(spec), Fs, _lib.ptr(coefs), 0, _lib.ptr(gains), df_state.bands_handle, B, Tf, F, nd, O, la, 0.0,
                                          0.0, _lib.ptr(out), Fs, _lib.stream()))

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(["dfx_k_df_apply"])
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    ms, n = _lib.prof_read().get("dfx_k_df_apply", (0.0, 0))
    _lib.prof_enable(None)
    return ms, n


def bench_host_io(model, df_state, x, steps: int) -> dict:
    """enhance() host to host: page-locked input and output batches, the upload of batch k+1 and the download of batch k-1 on copy streams
    of their own while batch k is computed (device input and output double-buffered).  The copies carry NO device-side dependency on the
    compute queues and nothing waits on them on the device: the host waits for the device at the top of every iteration (pass k-1, the
    download of k-2 and the upload of k are over — it would wait for pass k-1 inside enhance() anyway) and then enqueues the two copies and
    the pass.  (Copies that have to wait for a kernel's signal, or that are enqueued in the middle of a pass, were measured at 20-35 ms
    per step instead of 16-18: tools/dev/hostio_probe.py --variants, profiles/r03_hostio_probe.log.)
    Twice: float32 samples (what df.enhance.enhance() is handed) and 16-bit PCM (what the reference's file loop decodes and encodes around
    it, enhance.py:73-89 / io.py:25-84: dfx_enhance_pcm16 converts inside the STFT / ISTFT kernels, half the bytes over the link)."""
    from deepfilternet_amd.enhance import enhance
    from deepfilternet_amd.io import float_to_pcm16

    B, T = x.shape

    def pipeline(xsrc, copies_behind_front=True):
        nbytes = xsrc.numel() * xsrc.element_size()
        xh = torch.empty(xsrc.shape, dtype=xsrc.dtype, pin_memory=True)
        xh.copy_(xsrc)
        yh = [torch.empty(xsrc.shape, dtype=xsrc.dtype, pin_memory=True) for _ in range(2)]
        xd = [torch.empty_like(xsrc) for _ in range(2)]
        ys = [None, None]
        s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()

        def run(n):
            torch.cuda.synchronize()
            with torch.cuda.stream(s_in):
                xd[0].copy_(xh, non_blocking=True)
            def copies(k):
                if k >= 1:
                    with torch.cuda.stream(s_out):      # batch k-1 leaves under the compute of batch k (a consumer would take yh[(k-2) & 1] here)
                        yh[(k - 1) & 1].copy_(ys[(k - 1) & 1], non_blocking=True)
                if k + 1 < n:
                    with torch.cuda.stream(s_in):       # batch k+1 arrives under the compute of batch k
                        xd[(k + 1) & 1].copy_(xh, non_blocking=True)

            for k in range(n):
                torch.cuda.synchronize()                # pass k-1, the download of batch k-2 and the upload of batch k are over
                if not copies_behind_front:
                    copies(k)
                ys[k & 1] = enhance(model, df_state, xd[k & 1])
                if copies_behind_front:                 # enhance() returns when the encoder front has run and the GRU phase is enqueued (the engine's
                    copies(k)                           # staged enqueue): the two copies then run under the phase, not beside the HBM-bound front
            torch.cuda.synchronize()
            with torch.cuda.stream(s_out):
                yh[(n - 1) & 1].copy_(ys[(n - 1) & 1], non_blocking=True)
            torch.cuda.synchronize()

        run(3)
        t0 = time.perf_counter()
        run(steps)
        dt = (time.perf_counter() - t0) / steps
        out = yh[(steps - 1) & 1]
        ok = bool(torch.isfinite(out).all()) if out.dtype.is_floating_point else bool(int(out.abs().max()) > 0)
        return dt, nbytes, ok

    dt, nbytes, ok = pipeline(x, copies_behind_front=False)   # (f32 samples: 2 x 19.7 ms of DMA per step, longer than the pass: started as early as possible)
    x16 = float_to_pcm16(x)
    dt16, nbytes16, ok16 = pipeline(x16)
    dt16_before, _, _ = pipeline(x16, copies_behind_front=False)   # (the round-4 order: copies enqueued in front of the pass)
    # the resident step of this process, for the ratio (same loop as the headline, inputs in HBM)
    for _ in range(2):
        enhance(model, df_state, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        enhance(model, df_state, x)
    torch.cuda.synchronize()
    dt_res = (time.perf_counter() - t0) / steps
    return {"ms_per_step_host_to_host": dt * 1e3, "frames_per_s_host_to_host": B * (T // HOP) / dt,
            "pcie_gb_per_s_each_way": nbytes / dt / 1e9, "bytes_each_way_per_step": nbytes, "steps": steps, "finite": ok,
            "ms_per_step_host_to_host_pcm16": dt16 * 1e3, "frames_per_s_host_to_host_pcm16": B * (T // HOP) / dt16,
            "bytes_each_way_per_step_pcm16": nbytes16, "pcm16_nonzero": ok16,
            "ms_per_step_host_to_host_pcm16_copies_in_front_of_the_pass": dt16_before * 1e3,
            "ms_per_step_resident_same_process": dt_res * 1e3, "pcm16_over_resident": dt16 / dt_res, "f32_over_resident": dt / dt_res,
            "how": "page-locked [B, T] input and output; H2D of batch k+1 and D2H of batch k-1 on their own HIP streams under the compute of "
                   "batch k — 16-bit PCM: enqueued when enhance(k) has returned, i.e. behind its encoder front, under its GRU phase; f32: in front of the pass "
                   "(device input and output double-buffered, copies without device-side dependencies so that the DMA engines take them; the "
                   "last batch's download is inside the timed region); f32 samples, then 16-bit PCM samples (dfx_enhance_pcm16: the int16 <-> float "
                   "conversions of df/io.py inside the STFT loads / ISTFT stores); a process of its own; not part of `value`, which keeps its inputs "
                   "resident in HBM.  This box moves 57 GB/s in one direction and 2 x 28.7 GB/s in both at once (tools/dev/hostio_probe.py)"}


def bench_df_apply_o10(dev, df_state, B: int, Tf: int, iters: int = 20) -> dict:
    """BASELINE.json configs[4]: the deep filter of order 10 over nb_df = 96 bins (+ ERB gains on the rest) at batch 256, the kernel alone
    on the engine's own layout (rows of 488 bins = 64-byte aligned, tap-major coefficients); hipEvents on the launch stream."""
    from deepfilternet_amd import _lib

    F, Fs, nd, O, la, E = 481, 488, 96, 10, 3, 32
    g = torch.Generator(device=dev).manual_seed(0)
    spec = torch.randn((B, Tf, Fs, 2), device=dev, generator=g)
    coefs = 0.3 * torch.randn((B, O, Tf, nd, 2), device=dev, generator=g)
    gains = torch.rand((B, Tf, E), device=dev, generator=g)
    out = torch.empty_like(spec)
    L = _lib.lib()

    def run():
        _lib.check(L.dfx_df_apply_strided(_lib.ptr(spec), Fs, _lib.ptr(coefs), 0, _lib.ptr(gains), df_state.bands_handle, B, Tf, F, nd, O, la, 0.0,
                                          0.0, _lib.ptr(out), Fs, _lib.stream()))

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = (F * 8 + nd * O * 8 + E * 4 + F * 8) * B * Tf
    ach = nbytes / (ms * 1e-3) / 1e9
    return {"workload": f"deep filter order 10 (lookahead 3) + ERB gains, [{B}, {Tf}, 481] spectra, kernel alone", "kernel": "dfx_k_df_apply_rows<10>",
            "bound": "hbm", "avg_launch_ms": ms, "algorithmic_bytes_per_launch": nbytes, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "frames_per_s": B * Tf / (ms * 1e-3)}


if __name__ == "__main__":
    main()
