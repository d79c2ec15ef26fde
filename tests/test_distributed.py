"""The N>1 path: clips sharded over ranks, no data-path collective, one final gather (world_size 2, gloo, CPU).
Each rank runs the real kernels on the SIMT interpreter build of libdfx; the result must equal the single-process batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepfilternet_amd.distributed import shard_range


def test_shard_range_partitions_everything():
    for n in (0, 1, 5, 8, 255, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, x, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deepfilternet_amd import _lib
        from deepfilternet_amd.distributed import enhance_sharded
        from deepfilternet_amd.enhance import init_df
        from tests.helpers import named_params
        from tests.hipemu.build_emu import build

        _lib.use_library(build())
        p = named_params("defaults")
        model, df_state, _, _ = init_df(params=p, epoch="none", seed=3)
        xt = torch.from_numpy(x)
        h = enhance_sharded(model, df_state, xt)          # full batch in, this rank touches only its slice
        full = h.wait()
        lo, hi = shard_range(x.shape[0], rank, world)
        h2 = enhance_sharded(model, df_state, xt[lo:hi], presharded=True)  # ragged: 2 clips on rank 0, 1 on rank 1
        full2 = h2.wait()
        if rank == 0:
            assert full is not None and full2 is not None and torch.equal(full, full2)
            np.save(out_path, full.numpy())
        else:
            assert full is None and full2 is None
        assert h.local.shape[0] == hi - lo
        # the host-consumer form (no collective, every rank hands its own slice to the host): the same rows
        from deepfilternet_amd.distributed import check_distinct_devices, gather_device_ids

        hh = enhance_sharded(model, df_state, xt, gather="host")
        mine = hh.wait()
        assert torch.equal(mine, h.local) and mine.shape[0] == hi - lo
        np.save(out_path + f".host{rank}.npy", mine.numpy())
        ids = gather_device_ids()
        assert ids == [f"cpu:{r}" for r in range(world)]
        check_distinct_devices(ids)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_enhance_matches_single_process(tmp_path):
    from deepfilternet_amd import _lib
    from deepfilternet_amd.enhance import enhance, init_df
    from tests.helpers import named_params
    from tests.hipemu.build_emu import build

    rng = np.random.default_rng(11)
    x = (0.1 * rng.standard_normal((3, 480 * 5 + 17))).astype(np.float32)
    out_path = str(tmp_path / "gathered.npy")
    build()  # build once in the parent so the children only dlopen
    mp.spawn(_worker, args=(2, _free_port(), x, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    _lib.use_library(build())
    model, df_state, _, _ = init_df(params=named_params("defaults"), epoch="none", seed=3)
    ref = enhance(model, df_state, torch.from_numpy(x)).numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)  # rows are independent: sharding must not change a bit
    host = np.concatenate([np.load(out_path + f".host{r}.npy") for r in range(2)])
    assert np.array_equal(host, ref)                             # the host-consumer form: the ranks' own slices, no gather


def test_two_ranks_on_one_device_are_refused():
    from deepfilternet_amd.distributed import WorldError, check_distinct_devices

    check_distinct_devices(["GPU-a", "GPU-b", "GPU-c"])
    with pytest.raises(WorldError, match="ranks 0 and 2 are both on device GPU-a: one process per GPU"):
        check_distinct_devices(["GPU-a", "GPU-b", "GPU-a"])


# ---- the launcher bench.py --gpus N uses (deepfilternet_amd.distributed.check_world / launch_ranks / init_world)
def test_check_world_refuses_what_it_cannot_give(monkeypatch):
    from deepfilternet_amd.distributed import WorldError, check_world

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert check_world(1, 1) is None and check_world(2, 8) is None and check_world(2, None) is None
    with pytest.raises(WorldError, match="2 ranks requested, 1 device visible"):
        check_world(2, 1)
    with pytest.raises(WorldError, match="at least one"):
        check_world(0, 8)
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert check_world(4, 8) == (4, 3, 3)
    with pytest.raises(WorldError, match="--gpus 2 but the launcher started WORLD_SIZE=4"):
        check_world(2, 8)
    with pytest.raises(WorldError, match="4 ranks requested, 2 devices"):
        check_world(4, 2)


def _run(cmd, env=None, timeout=900):
    import subprocess
    import sys

    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable] + cmd, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_starts_its_own_ranks(tmp_path):
    """`script --gpus 2` with no launcher in the environment starts 2 ranks itself (gloo, interpreter build); the gathered result is
    the single-process one, bit for bit."""
    import json

    from tests.hipemu.build_emu import build

    build()
    here = os.path.dirname(os.path.abspath(__file__))
    script = os.path.join(here, "rank_script.py")
    o1, o2 = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    r2 = _run([script, "--gpus", "2", "--out", o2])
    assert r2.returncode == 0, r2.stderr[-2000:]
    line = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"n_gpus": 2, "ranks_in_group": 2, "clips": 3}
    r1 = _run([script, "--gpus", "1", "--out", o1])
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert np.array_equal(np.load(o1), np.load(o2))


def test_gpus_flag_fails_loudly():
    here = os.path.dirname(os.path.abspath(__file__))
    script = os.path.join(here, "rank_script.py")
    r = _run([script, "--gpus", "2", "--devices", "1"])
    assert r.returncode != 0 and "2 ranks requested, 1 device visible" in r.stderr
    r = _run([script, "--gpus", "2"], env={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but the launcher started WORLD_SIZE=3" in r.stderr
    # bench.py itself: the mismatch is refused before anything touches a device
    r = _run([os.path.join(os.path.dirname(here), "bench.py"), "--gpus", "2"], env={"WORLD_SIZE": "8", "RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but the launcher started WORLD_SIZE=8" in r.stderr


def test_launch_ranks_ends_the_job_when_a_rank_fails(tmp_path):
    import sys

    from deepfilternet_amd.distributed import launch_ranks

    s = tmp_path / "r.py"
    s.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    import time

    t0 = time.monotonic()
    assert launch_ranks([sys.executable, str(s)], 2) == 7
    assert time.monotonic() - t0 < 30
