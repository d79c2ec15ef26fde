"""The reference's C API (libDF/src/capi.rs:83-253: df_create / df_get_frame_length / df_next_log_msg / df_set_atten_lim /
df_set_post_filter_beta / df_process_frame / df_free) served by libdfx.so (include/df_capi.h), called through ctypes exactly as a C
host would call it: host float buffers of one hop, one mono stream per state.  Checked against the streaming oracle (the
reference loop's logic with its default thresholds) and against the batched runtime."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import stream_oracle as S
from tests.helpers import emu_subset, named_params, rms, torch_sd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOP = 480
CAPI_THRESHOLDS = (-15.0, 35.0, 35.0)   # DFState::new, capi.rs:27-34


def _capi(lib):
    fp = C.POINTER(C.c_float)
    lib.df_create.restype, lib.df_create.argtypes = C.c_void_p, [C.c_char_p, C.c_float, C.c_char_p]
    lib.df_get_frame_length.restype, lib.df_get_frame_length.argtypes = C.c_size_t, [C.c_void_p]
    lib.df_next_log_msg.restype, lib.df_next_log_msg.argtypes = C.c_void_p, [C.c_void_p]
    lib.df_free_log_msg.restype, lib.df_free_log_msg.argtypes = None, [C.c_void_p]
    lib.df_set_atten_lim.restype, lib.df_set_atten_lim.argtypes = None, [C.c_void_p, C.c_float]
    lib.df_set_post_filter_beta.restype, lib.df_set_post_filter_beta.argtypes = None, [C.c_void_p, C.c_float]
    lib.df_process_frame.restype, lib.df_process_frame.argtypes = C.c_float, [C.c_void_p, fp, fp]
    lib.df_free.restype, lib.df_free.argtypes = None, [C.c_void_p]
    return lib


def test_header_symbols_exported():
    from deepfilternet_amd.build import build

    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "df_capi.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(df_[a-z_]+)\s*\(", src)))
    assert names == ["df_create", "df_free", "df_free_log_msg", "df_get_frame_length", "df_next_log_msg", "df_process_frame",
                     "df_process_frame_raw", "df_set_atten_lim", "df_set_post_filter_beta"]
    lib = C.CDLL(build())
    for n in names:
        assert hasattr(lib, n), n
    lib = _capi(lib)
    assert lib.df_create(b"/nonexistent/model.dfx", 100.0, None) is None      # no model, no state (the reference panics here)


def _process(lib, st, x):
    fp = C.POINTER(C.c_float)
    y = np.zeros_like(x)
    lsnr = np.zeros(len(x) // HOP, np.float32)
    for k in range(len(x) // HOP):
        xin = np.ascontiguousarray(x[k * HOP:(k + 1) * HOP])
        out = np.zeros(HOP, np.float32)
        lsnr[k] = lib.df_process_frame(st, xin.ctypes.data_as(fp), out.ctypes.data_as(fp))
        y[k * HOP:(k + 1) * HOP] = out
    return y, lsnr


def test_df_capi_frame_loop(backend, tmp_path):
    from deepfilternet_amd import _lib, export_dfx
    from deepfilternet_amd.state_dict import random_state_dict

    p = named_params("defaults" if backend == "emu" else "pf32")   # the interpreter is slow: the small model there
    sd_np = random_state_dict(p, 9)
    path = export_dfx(str(tmp_path / "model.dfx"), params=p, state_dict=sd_np)
    lib = _capi(C.CDLL(_lib.library_path()))
    st = lib.df_create(os.fsencode(path), 100.0, b"info")
    assert st, lib.dfx_last_error
    assert lib.df_get_frame_length(st) == HOP
    msgs = []
    while True:
        m = lib.df_next_log_msg(st)
        if not m:
            break
        msgs.append(C.cast(m, C.c_char_p).value.decode())
        lib.df_free_log_msg(m)
    assert any(f"lookahead {p.df_lookahead}" in m for m in msgs), msgs
    T = (3 if emu_subset(backend) else 5) if backend == "emu" else 20
    rng = np.random.default_rng(7)
    x = (0.1 * rng.standard_normal(HOP * T)).astype(np.float32)
    sd = torch_sd(p, 9)
    # what the reference's C API fixes in DFState::new (capi.rs:27-34): thresholds -15 / 35 / 35 dB, post filter off
    y, lsnr = _process(lib, st, x)
    yr, lr, _ = S.process_stream(p, sd, x, pf_beta=0.0, thresholds=CAPI_THRESHOLDS)
    assert rms(y - yr) < 1e-6 and np.abs(lsnr - lr)[p.df_lookahead:].max() < 1e-3
    lib.df_free(st)
    # post filter + attenuation limit through the setters, fresh state
    st = lib.df_create(os.fsencode(path), 100.0, None)
    assert not lib.df_next_log_msg(st)                       # no log level, no messages (capi.rs:91-101)
    lib.df_set_post_filter_beta(st, 0.02)
    lib.df_set_atten_lim(st, 12.0)
    y, _ = _process(lib, st, x)
    yr, _, _ = S.process_stream(p, sd, x, atten_lim_db=12.0, pf_beta=0.02, thresholds=CAPI_THRESHOLDS)
    assert rms(y - yr) < 1e-6
    lib.df_free(st)
    # a corrupt file is refused
    bad = tmp_path / "bad.dfx"
    bad.write_bytes(open(path, "rb").read()[:1000])
    assert lib.df_create(os.fsencode(str(bad)), 100.0, None) is None


def test_process_frame_raw(backend, tmp_path):
    """df_process_frame_raw (capi.rs:172-210 -> DfTract::process_raw, tract.rs:441-507): spectral frames in, raw ERB gains and DF
    coefficients out, NULL where the stage decision skipped the decoder — through the batched dfx_stream_process_raw (two streams) and
    through the reference-named C entry point, against oracle.stream_oracle.process_raw_frames."""
    from deepfilternet_amd import _lib, export_dfx
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.state_dict import random_state_dict
    from deepfilternet_amd.streaming import DfStream
    from oracle import libdf_oracle as L

    if emu_subset(backend):
        pytest.skip("interpreter subset: runs on the GPU (DFX_EMU_ALL=1 runs it on the interpreter too)")
    p = named_params("pf32")
    sd_np = random_state_dict(p, 9)
    sd = torch_sd(p, 9)
    K = 8 if backend == "emu" else 24
    rng = np.random.default_rng(11)
    x = (0.1 * rng.standard_normal((2, HOP * K))).astype(np.float32)
    x[1] *= np.linspace(0.02, 2.0, HOP * K).astype(np.float32)
    spec = L.DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs).analysis(x)        # [2, K, F] complex64
    # thresholds inside this model's lsnr range (stage 1 always, stage 2 for the lower half), away from every observed value
    free = [S.process_raw_frames(p, sd, spec[i], thresholds=(-1e9, 1e9, 1e9)) for i in range(2)]
    vals = np.sort([r[0] for f in free for r in f if r[0] is not None])
    j = max(range(len(vals) // 2 - 2, len(vals) // 2 + 2), key=lambda i: vals[i + 1] - vals[i])
    thr = (-1e9, 1e9, float(vals[j] + vals[j + 1]) / 2)
    ref = [S.process_raw_frames(p, sd, spec[i], thresholds=thr) for i in range(2)]
    assert any(r[2] is None and r[1] is not None for f in ref for r in f) and any(r[2] is not None for f in ref for r in f)
    model, df_state, _, _ = init_df(params=p, state_dict=sd_np, epoch="none")
    rt = DfStream(model, df_state, streams=2, gating=True, thresholds=thr)
    for k in range(K):
        lsnr, gains, coefs, stages = rt.process_raw(torch.from_numpy(np.ascontiguousarray(spec[:, k])))
        for i in range(2):
            rl, rg, rc = ref[i][k]
            if rl is None:
                assert int(stages[i]) == 0 and float(lsnr[i]) == -15.0
                continue
            assert abs(float(lsnr[i]) - rl) < 1e-3
            assert bool(int(stages[i]) & 2) == (rg is not None) and bool(int(stages[i]) & 8) == (rc is not None)
            if rg is not None:
                assert np.abs(gains[i].numpy() - rg).max() < 1e-5
            if rc is not None:
                assert np.abs(coefs[i].numpy() - rc).max() < 1e-5
    # lsnr below min_db_thresh: the reference returns Some(zeros) as gains and no coefficients (tract.rs:485-486, :658-661)
    rz = DfStream(model, df_state, streams=2, gating=True, thresholds=(1e9, 2e9, 2e9))
    for k in range(3 + p.df_lookahead):
        lsnr, gains, coefs, stages = rz.process_raw(torch.from_numpy(np.ascontiguousarray(spec[:, k])))
        if k >= p.df_lookahead:
            assert np.all(stages.numpy() == 2) and np.all(gains.numpy() == 0)
    # the reference-named entry point, one stream: the C API's thresholds (-15 / 35 / 35 dB, capi.rs:27-34) -> both stages on these signals
    path = export_dfx(str(tmp_path / "model.dfx"), params=p, state_dict=sd_np)
    lib = _capi(C.CDLL(_lib.library_path()))
    fp = C.POINTER(C.c_float)
    lib.df_process_frame_raw.restype = C.c_float
    lib.df_process_frame_raw.argtypes = [C.c_void_p, fp, C.POINTER(fp), C.POINTER(fp)]
    st = lib.df_create(os.fsencode(path), 100.0, None)
    assert st
    dflt = S.process_raw_frames(p, sd, spec[0], thresholds=CAPI_THRESHOLDS)
    for k in range(min(K, 6)):
        frame = np.ascontiguousarray(spec[0, k]).view(np.float32).copy()
        g = np.zeros(p.nb_erb, np.float32)
        c = np.zeros((p.df_order, p.nb_df, 2), np.float32)
        gp_, cp_ = g.ctypes.data_as(fp), c.ctypes.data_as(fp)
        lsnr = lib.df_process_frame_raw(st, frame.ctypes.data_as(fp), C.byref(gp_), C.byref(cp_))
        rl, rg, rc = dflt[k]
        if rl is None:
            assert not gp_ and not cp_ and lsnr == -15.0
            continue
        assert abs(lsnr - rl) < 1e-3 and bool(gp_) == (rg is not None) and bool(cp_) == (rc is not None)
        assert np.abs(g - rg).max() < 1e-5 and np.abs(c.view(np.complex64)[..., 0] - rc).max() < 1e-5
    lib.df_free(st)


def test_dfx_file_versions(backend, tmp_path):
    """A version-1 .dfx file (before dfx_model_cfg grew emb_gru_skip_enc / emb_gru_skip / enc_concat) is a version-2 file with those three
    fields absent = 0 and still loads; an unknown version is refused with a message that says what to do."""
    import ctypes as C
    import struct

    from deepfilternet_amd import _lib, export_dfx
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.state_dict import random_state_dict

    p = ModelParams.defaults()
    sd = random_state_dict(p, 4)
    v2 = export_dfx(str(tmp_path / "v2.dfx"), params=p, state_dict=sd)
    raw = open(v2, "rb").read()
    magic, ver, csz = raw[:4], *struct.unpack("<II", raw[4:12])
    assert magic == b"DFXM" and ver == 2 and csz == C.sizeof(_lib.ModelCfg)
    cfg, rest = raw[12:12 + csz], raw[12 + csz:]
    assert cfg[-12:] == b"\0" * 12           # the defaults: no embedding-GRU skips, no concat
    v1 = str(tmp_path / "v1.dfx")
    open(v1, "wb").write(magic + struct.pack("<II", 1, csz - 12) + cfg[:-12] + rest)
    L = _lib.lib()
    for path in (v2, v1):
        h = C.c_void_p()
        _lib.check(L.dfx_model_load_file(os.fsencode(path), C.byref(h)))
        got = _lib.ModelCfg()
        _lib.check(L.dfx_model_cfg_get(h, C.byref(got)))
        assert bytes(got) == cfg, path
        L.dfx_model_free(h)
    v9 = str(tmp_path / "v9.dfx")
    open(v9, "wb").write(magic + struct.pack("<II", 9, csz) + cfg + rest)
    h = C.c_void_p()
    assert L.dfx_model_load_file(os.fsencode(v9), C.byref(h)) != 0
    assert b"version 9" in L.dfx_last_error() and b"re-export" in L.dfx_last_error()


def test_process_and_process_raw_share_one_state(backend):
    """A handle may be driven through the waveform entry (dfx_stream_process) and through the spectral one (dfx_stream_process_raw) in
    turns: the waveform entry keeps the windows of a gated handle in linear buffers, the spectral one in ring form — whichever form holds
    the history, the other entry continues from it.  Handle A takes every hop as spectra; handle B takes the first hops as waveform and
    the rest as spectra: the raw answers of the last hops agree."""
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.state_dict import random_state_dict
    from deepfilternet_amd.streaming import DfStream
    from oracle import libdf_oracle as L

    if emu_subset(backend):
        pytest.skip("interpreter subset: runs on the GPU (DFX_EMU_ALL=1 runs it on the interpreter too)")
    p = named_params("pf32")
    sd_np = random_state_dict(p, 9)
    K, K0 = (8, 5) if backend == "emu" else (20, 13)
    rng = np.random.default_rng(12)
    x = (0.1 * rng.standard_normal((2, HOP * K))).astype(np.float32)
    spec = L.DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs).analysis(x)        # [2, K, F] complex64
    model, df_state, _, _ = init_df(params=p, state_dict=sd_np, epoch="none")
    thr = (-1e9, 1e9, 1e9)   # every stage runs on every hop: both entries then advance the same state
    ra = DfStream(model, df_state, streams=2, gating=True, thresholds=thr)
    rb = DfStream(model, df_state, streams=2, gating=True, thresholds=thr)
    outs_a = [ra.process_raw(torch.from_numpy(np.ascontiguousarray(spec[:, k]))) for k in range(K)]
    for k in range(K0):
        rb.process(torch.from_numpy(x[:, k * HOP:(k + 1) * HOP]))
    for k in range(K0, K):
        lsnr, gains, coefs, stages = rb.process_raw(torch.from_numpy(np.ascontiguousarray(spec[:, k])))
        la, ga, ca, sa = outs_a[k]
        assert np.array_equal(stages.numpy(), sa.numpy()) and np.all(stages.numpy() & 2)
        assert np.abs(lsnr.numpy() - la.numpy()).max() < 1e-3
        assert np.abs(gains.numpy() - ga.numpy()).max() < 1e-5, k
        assert np.abs(coefs.numpy() - ca.numpy()).max() < 1e-5, k
    model.check()
