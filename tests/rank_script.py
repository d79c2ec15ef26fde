"""Helper of tests/test_distributed.py (not a test): one rank of a job started by deepfilternet_amd.distributed.launch_ranks or by
torch.distributed.run.  Mirrors bench.py's rank handling — check_world(--gpus) -> (launch the ranks | take the launcher's) ->
init_world -> enhance_sharded — on the CPU interpreter build with gloo.  Rank 0 prints one JSON line."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--devices", type=int, default=-1, help="pretend this many devices are visible (-1: not a GPU run)")
    args = ap.parse_args()
    from deepfilternet_amd.distributed import WorldError, check_world, enhance_sharded, init_world, launch_ranks

    try:
        env_world = check_world(args.gpus, None if args.devices < 0 else args.devices)
    except WorldError as e:
        raise SystemExit(f"rank_script: {e}")
    if env_world is None and args.gpus > 1:
        raise SystemExit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=600))
    world, rank, _ = env_world if env_world is not None else (1, 0, 0)
    import torch.distributed as dist

    if world > 1:
        init_world("gloo", world, rank)
    from deepfilternet_amd import _lib
    from deepfilternet_amd.enhance import init_df
    from tests.helpers import named_params
    from tests.hipemu.build_emu import build

    _lib.use_library(build())
    model, df_state, _, _ = init_df(params=named_params("defaults"), epoch="none", seed=3)
    rng = np.random.default_rng(11)
    x = torch.from_numpy((0.1 * rng.standard_normal((3, 480 * 4 + 5))).astype(np.float32))
    full = enhance_sharded(model, df_state, x).wait()
    if rank == 0:
        if args.out:
            np.save(args.out, full.numpy())
        print(json.dumps({"n_gpus": world, "ranks_in_group": dist.get_world_size() if world > 1 else 1, "clips": int(full.shape[0])}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
