"""Passes that meet on one GPU: two handles, two threads, two processes (round-5 review, item 1; docs/measurements.md R6.1).

What went wrong in round 5 ("two handles return wrong samples now and then") was not the GRU phase: one handle's STFT kernels ran beside the
other handle's fp16-split kernels, and on this MI355X a wave's packed fp32 operations miscompute while another wave of its SIMD executes a
double-rate matrix operation.  The library carries no packed fp32 operations any more (tests/test_isa.py); these tests run the situations that
used to fail — at rates of 5 ... 25 % per pass — and hold every result to the bits of a pass made alone.
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

from tests.helpers import named_params

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _noise(B, T, seed):
    return torch.from_numpy((0.1 * np.random.default_rng(seed).standard_normal((B, T))).astype(np.float32))


def test_stft_kernels_beside_another_handles_passes(hip_backend):
    """The strongest form of the old failure: df_features (dfx_k_analysis + dfx_k_norm_scan4) in a loop on one stream while another handle runs
    whole passes on another.  With packed fp32 operations in the STFT kernel 20-25 % of the iterations came back wrong."""
    from deepfilternet_amd.enhance import df_features, enhance, init_df
    from deepfilternet_amd.libdf import DF

    p = named_params("df3")
    st = DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs)
    x = _noise(256, 96960, 3).cuda()
    ref = [t.clone() for t in df_features(x, st, p.nb_df)]
    torch.cuda.synchronize()
    model, mst = init_df(params=p, epoch="none", seed=4)[:2]
    xo = _noise(256, 96000, 5).cuda()
    stop, errs, passes = threading.Event(), [], [0]

    def other():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                while not stop.is_set():
                    enhance(model, mst, xo)
                    passes[0] += 1
                    if passes[0] % 4 == 0:
                        torch.cuda.current_stream().synchronize()
            model.check()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    import time

    t = threading.Thread(target=other)
    t.start()
    t0 = time.time()
    while passes[0] < 2 and not errs and time.time() - t0 < 120:   # (the other handle's first pass carries its handshake and allocations)
        time.sleep(0.01)
    wrong = n = 0
    p0 = passes[0]
    with torch.cuda.stream(torch.cuda.Stream()):
        while (n < 300 or passes[0] - p0 < 8) and n < 5000 and not errs:
            got = df_features(x, st, p.nb_df)
            wrong += int(not all(torch.equal(a, b) for a, b in zip(got, ref)))
            n += 1
    stop.set()
    t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert passes[0] - p0 >= 8, (passes, n)   # the other handle really ran beside the loop
    assert wrong == 0, f"{wrong} of {n} STFT results differ from the result computed alone"


def test_two_handles_share_the_process_streams_and_both_run_the_persistent_phase(hip_backend):
    """The internal streams belong to the process (DfxLaneSet): a second handle neither runs out of hardware queues nor is put on the
    event-synchronised form by its handshake, and its passes carry the bits of the first handle's."""
    from deepfilternet_amd.enhance import enhance, init_df

    p = named_params("df3")
    x = _noise(64, 48000, 9).cuda()
    models = [init_df(params=p, epoch="none", seed=4)[:2] for _ in range(3)]
    ys = [enhance(m, s, x).cpu() for m, s in models]
    for m, _ in models:
        m.check()
        assert m.query(m.Q_HWQ_PROBE) == models[0][0].query(models[0][0].Q_HWQ_PROBE)
        assert m.query(m.Q_GRU_PERSISTENT) == models[0][0].query(models[0][0].Q_GRU_PERSISTENT)
    assert all(torch.equal(y, ys[0]) for y in ys)


_WORKER = r"""
import os, sys, time
sys.path.insert(0, {repo!r})
import numpy as np, torch
from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.enhance import enhance, init_df
from deepfilternet_amd.state_dict import random_state_dict
tag, go, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
p = ModelParams.deepfilternet3()
model, st = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")[:2]
x = torch.from_numpy((0.1 * np.random.default_rng(7).standard_normal((256, 96000))).astype(np.float32)).cuda()
ref = enhance(model, st, x).clone()
torch.cuda.synchronize(); model.check()
open(go + "." + tag + ".ready", "w").close()
t0 = time.time()
while not os.path.exists(go):
    if time.time() - t0 > 120: sys.exit(3)
    time.sleep(0.005)
wrong = 0
for i in range(n):
    y = enhance(model, st, x)
    wrong += int(not torch.equal(y, ref))
torch.cuda.synchronize(); model.check()
print("RESULT", tag, "wrong", wrong, "of", n, "persistent", model.query(5), "ticket_busy", model.query(6), flush=True)
sys.exit(1 if wrong else 0)
"""


def test_two_processes_on_one_device(hip_backend, tmp_path):
    """Two PROCESSES enhance on the same GPU at the same time.  Both get right samples and neither raises a fault: a persistent GRU phase needs all
    its workgroups resident, so a process only starts one while it holds the device's ticket (/dev/shm/dfx_persistent_<bus>.lock) and runs the
    event-synchronised form of the phase when the other process has it (dfx_model_query DFX_Q_PASSES_TICKET_BUSY counts those)."""
    go = str(tmp_path / "go")
    src = _WORKER.format(repo=REPO)
    env = dict(os.environ, PYTHONPATH=REPO, DFX_QUIET="1")
    procs = [subprocess.Popen([sys.executable, "-c", src, tag, go, "40"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for tag in ("a", "b")]
    import time

    t0 = time.time()
    while not all(os.path.exists(f"{go}.{tag}.ready") for tag in ("a", "b")):
        assert time.time() - t0 < 300 and all(pr.poll() is None for pr in procs), [pr.communicate()[0][-2000:] for pr in procs if pr.poll() is not None]
        time.sleep(0.05)
    open(go, "w").close()
    outs = [pr.communicate(timeout=600)[0] for pr in procs]
    for pr, out in zip(procs, outs):
        assert pr.returncode == 0, out[-3000:]
    res = [line for out in outs for line in out.splitlines() if line.startswith("RESULT")]
    assert len(res) == 2, outs
    print("\n".join(res))
    counts = [dict(zip(r.split()[2::2], r.split()[3::2])) for r in res]
    assert all(int(c["wrong"]) == 0 for c in counts), res
    # every big pass ran one form or the other
    assert all(int(c["persistent"]) + int(c["ticket_busy"]) >= 40 for c in counts), res


def test_a_faulted_pass_hands_back_nan(hip_backend):
    """A pass in which a kernel raised a fault does not return plausible samples: the finishing kernel reads the model's error words and stores
    NaN (16-bit PCM: zeros); the host is told by the words as before."""
    import ctypes as C

    from deepfilternet_amd import _lib
    from tests.test_faults import _overflowing_model

    model, df_state = _overflowing_model()
    L = _lib.lib()
    for dtype, fn in ((torch.float32, L.dfx_enhance), (torch.int16, L.dfx_enhance_pcm16)):
        x = _noise(3, 480 * 40, 0)
        x = (x * 32768).to(torch.int16).cuda() if dtype == torch.int16 else x.cuda()
        y = torch.full(x.shape, 7, dtype=dtype, device=x.device)
        n = C.c_int64()
        _lib.check(L.dfx_enhance_workspace_bytes(model.handle, df_state.handle, x.shape[0], x.shape[1], 1, C.byref(n)))
        ws = model.workspace(n.value)
        _lib.check(fn(model.handle, df_state.handle, _lib.ptr(x), x.shape[0], x.shape[1], 1, 0.0, _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.stream()))
        torch.cuda.synchronize()
        if dtype == torch.int16:
            assert int(y.abs().max()) == 0, "a faulted pass returned 16-bit samples other than silence"
        else:
            assert torch.isnan(y).all(), "a faulted pass returned finite samples"
        with pytest.raises(_lib.DfxError, match="fp16-split"):
            model.check()
