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
