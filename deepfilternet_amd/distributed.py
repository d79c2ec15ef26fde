"""Multi-GPU sharding of ``enhance()``: clips are independent (STFT memories, norm states and GRU h0 are
per clip, BatchNorm is in eval mode), so the batch is split across the ranks of one node — one process per GPU — with NO
collective on the data path.  The only exchange is the optional final gather of the finished waveforms over RCCL/xGMI
(``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests), which ``enhance_sharded`` issues asynchronously so
that it overlaps the next batch.

The reference has no distributed code at all; this module is the engine's addition above the drop-in
boundary and keeps ``enhance``'s argument meaning: ``audio`` is the full ``[C, T]`` batch (or this rank's slice).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of ``n`` clips owned by ``rank``: sizes differ by at most one, lower ranks get the extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GatherHandle:
    """Result of ``enhance_sharded(..., gather=True)``: ``wait()`` returns the full ``[C, T]`` batch on ``dst`` (None elsewhere)."""

    def __init__(self, work, parts: Optional[List[torch.Tensor]], sizes: List[int], local: torch.Tensor):
        self._work, self._parts, self._sizes, self.local = work, parts, sizes, local

    def wait(self) -> Optional[torch.Tensor]:
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._parts is None:
            return None
        return torch.cat([p[:n] for p, n in zip(self._parts, self._sizes)], dim=0)


class HostHandle:
    """Result of ``enhance_sharded(..., gather="host")``: the consumer is the host, so nothing crosses xGMI — every rank copies its own slice
    to page-locked host memory on its own stream ("if the consumer is the host, skip the gather").  ``wait()`` returns this rank's
    ``[n, T]`` host tensor; ``local`` is the device result."""

    def __init__(self, host: torch.Tensor, event, local: torch.Tensor):
        self._host, self._event, self.local = host, event, local

    def wait(self) -> torch.Tensor:
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        return self._host


def check_distinct_devices(ids: List[str]) -> None:
    """Every rank of a node must sit on a GPU of its own.  Two ranks on one device would run two passes of the persistent GRU phase side by
    side — each needs all its ~160 workgroups resident, so they fall back to taking turns through the device's ticket at best (half the
    throughput, reported as scaling) — and is always a launcher mistake (LOCAL_RANK not applied, HIP_VISIBLE_DEVICES narrowed).  ``ids`` = one
    device identity per rank (PCI bus id / uuid)."""
    seen = {}
    for r, d in enumerate(ids):
        if d in seen:
            raise WorldError(f"ranks {seen[d]} and {r} are both on device {d}: one process per GPU (check LOCAL_RANK / HIP_VISIBLE_DEVICES of the "
                             "launcher); two ranks on one device would share its CUs between two persistent GRU phases (docs/measurements.md R6.1)")
        seen[d] = r


def gather_device_ids(device=None, group=None) -> List[str]:
    """The device identity of every rank (all_gather_object; a CPU run reports ``cpu:<rank>``)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if device is not None and torch.cuda.is_available():
        p = torch.cuda.get_device_properties(device)
        mine = str(getattr(p, "uuid", None) or getattr(p, "pci_bus_id", None) or f"{p.name}:{torch.device(device).index}")
        mine = f"{mine}:{getattr(p, 'pci_bus_id', '')}:{getattr(p, 'pci_device_id', '')}"
    else:
        mine = f"cpu:{rank}"
    if not dist.is_initialized():
        return [mine]
    out: List[Optional[str]] = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, mine, group=group)
    return [str(o) for o in out]


def enhance_sharded(model, df_state, audio: torch.Tensor, pad: bool = True, atten_lim_db: Optional[float] = None, *,
                    group=None, presharded: bool = False, counts: Optional[List[int]] = None, gather=True,
                    dst: int = 0, enhance_fn=None):
    """Enhance this rank's clips of ``audio`` and (optionally) gather every rank's output to ``dst``.

    audio       full batch ``[C, T]`` (every rank passes the same tensor; only its own slice is touched) or, with
                ``presharded=True``, this rank's slice only.
    counts      with ``presharded``: clips held by every rank, if known (saves the size exchange, which synchronises the host).
    gather      True: returns a :class:`GatherHandle` (asynchronous ``dist.gather`` to ``dst``); "host": no collective, every rank copies its slice to
                page-locked host memory (:class:`HostHandle`); False: returns the local output.
    enhance_fn  defaults to :func:`deepfilternet_amd.enhance.enhance` (hook for tests).
    """
    if enhance_fn is None:
        from .enhance import enhance as enhance_fn
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sizes: Optional[List[int]] = None
    if presharded:
        mine = audio
        if counts is not None:
            if len(counts) != world or counts[rank] != audio.shape[0]:
                raise ValueError("counts must list every rank's number of clips")
            sizes = [int(c) for c in counts]
    else:
        lo, hi = shard_range(audio.shape[0], rank, world)
        mine = audio[lo:hi]
        if mine.device.type == "cpu" and not mine.is_pinned() and torch.cuda.is_available() and mine.numel() >= (1 << 18):
            # this rank's slice of a pageable host batch: page-lock it (a copy of 1/world of the batch) so that enhance() moves it — and
            # the result — by DMA on its stream instead of through the driver's pageable staging
            mine = mine.pin_memory()
        sizes = [b - a for a, b in (shard_range(audio.shape[0], r, world) for r in range(world))]
    y = enhance_fn(model, df_state, mine, pad=pad, atten_lim_db=atten_lim_db)
    if gather == "host":
        if y.device.type != "cuda":
            return HostHandle(y, None, y)
        host = torch.empty(y.shape, dtype=y.dtype, pin_memory=True)
        host.copy_(y, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(y.device))
        return HostHandle(host, ev, y)
    if not gather or world == 1:
        return GatherHandle(None, [y], [y.shape[0]], y) if gather else y
    if sizes is None:  # ranks may hold different numbers of clips: exchange the counts first (tiny; blocks the host once)
        sizes_t = torch.zeros(world, dtype=torch.int64)
        sizes_t[rank] = y.shape[0]
        dev_sizes = sizes_t.to(y.device) if dist.get_backend(group) == "nccl" else sizes_t
        dist.all_reduce(dev_sizes, group=group)
        sizes = [int(v) for v in dev_sizes.cpu().tolist()]
    nmax = max(sizes)
    send = y
    if y.shape[0] < nmax:  # gather needs equal shapes: pad the short ranks (at most one clip)
        send = torch.zeros((nmax,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
        send[: y.shape[0]] = y
    parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    work = dist.gather(send.contiguous(), parts, dst=dst, group=group, async_op=True)
    return GatherHandle(work, parts, sizes, y)


# ------------------------------------------------------------------------------------------------ launcher
# One process per GPU.  `python bench.py --gpus N` (or any script that calls `ensure_world`) starts its own N ranks when no
# launcher did; under an external launcher (torch.distributed.run: WORLD_SIZE / RANK / LOCAL_RANK in the environment) it checks
# that the launcher's world is the one that was asked for.  Nothing here needs a GPU: the CPU tests drive it with gloo.
_RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
             "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT")


class WorldError(RuntimeError):
    """The requested number of ranks cannot be had (devices missing, or the external launcher started another number)."""


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def world_from_env() -> Optional[Tuple[int, int, int]]:
    """(world, rank, local_rank) when a launcher set them, else None."""
    import os

    if "WORLD_SIZE" not in os.environ:
        return None
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ.get("RANK", "0"))
    return world, rank, int(os.environ.get("LOCAL_RANK", str(rank)))


def check_world(requested: int, device_count: Optional[int]) -> Optional[Tuple[int, int, int]]:
    """Validates ``--gpus requested`` against the environment.  Returns the launcher's (world, rank, local_rank), or None when this
    process has to start the ranks itself.  ``device_count`` = visible GPUs (None: not a GPU run, e.g. the gloo tests)."""
    if requested < 1:
        raise WorldError(f"--gpus {requested}: need at least one rank")
    env = world_from_env()
    if env is not None and env[0] != requested:
        raise WorldError(f"--gpus {requested} but the launcher started WORLD_SIZE={env[0]} ranks: pass the same number to both "
                         f"(python -m torch.distributed.run --nproc-per-node {requested} ... --gpus {requested})")
    if device_count is not None and device_count < requested:
        raise WorldError(f"{requested} ranks requested, {device_count} device{'s' if device_count != 1 else ''} visible: "
                         "one process per GPU, ranks never share a device")
    if env is not None and not (0 <= env[1] < env[0]):
        raise WorldError(f"RANK={env[1]} outside WORLD_SIZE={env[0]}")
    return env


def launch_ranks(argv: List[str], nproc: int, *, extra_env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Starts ``nproc`` copies of ``argv`` (a full command line), one per rank, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set (rendezvous on 127.0.0.1, a free port), and waits for them.  Rank 0 inherits stdout (its one JSON line is the
    job's); the first failing rank ends the others.  Returns the job's exit code (0 only if every rank returned 0)."""
    import os
    import subprocess
    import time

    port = free_port()
    procs = []
    for r in range(nproc):
        env = {k: v for k, v in os.environ.items() if k not in _RANK_ENV}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this driver)
        env.update(extra_env or {})
        procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    t0, rc = time.monotonic(), 0
    live = set(range(nproc))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
            if rc == 0:
                rc = 124
            for r in live:       # exactly the processes started above
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        if live:
            time.sleep(0.05)
    return rc


def init_world(backend: str, world: int, rank: int, device=None) -> None:
    """``init_process_group`` on 127.0.0.1 + the check the bench relies on: the group really has ``world`` ranks."""
    import os

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    if dist.get_world_size() != world or dist.get_rank() != rank:
        raise WorldError(f"process group has {dist.get_world_size()} ranks (this is {dist.get_rank()}), expected {world} / {rank}")
