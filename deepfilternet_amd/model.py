"""DeepFilterNet3 on the HIP engine: the host-side mirror of ``df.deepfilternet3.DfNet`` (deepfilternet3.py:334-456).

``DfNet`` keeps the reference's call signature — ``model(spec, feat_erb, feat_spec) -> (spec_e, m, lsnr, df_coefs)`` with
the reference's tensor shapes — and a ``state_dict``-shaped constructor, but it is not an ``nn.Module`` that computes in
PyTorch: every FLOP happens in libdfx.so.  Weight loading mirrors checkpoint.py:46-103 (``read_cp``): plain
``torch.load`` of a state-dict, the legacy ``clc -> df`` key rename, buffers ``erb_fb``/``mask.erb_inv_fb`` ignored.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import re
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import ModelParams
from .libdf import DF

def make_cfg(p: ModelParams) -> _lib.ModelCfg:
    p.check_supported()
    return p.to_cfg()


def tensor_manifest(cfg: _lib.ModelCfg):
    """[(name, shape, offset)] as the C library wants them packed (dfx_model_tensor_info)."""
    L = _lib.lib()
    n = C.c_int()
    _lib.check(L.dfx_model_tensor_count(C.byref(cfg), C.byref(n)))
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    nd, off = C.c_int(), C.c_int64()
    for i in range(n.value):
        _lib.check(L.dfx_model_tensor_info(C.byref(cfg), i, name, 256, shape, C.byref(nd), C.byref(off)))
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), int(off.value)))
    return out


def pack_state_dict(cfg: _lib.ModelCfg, sd: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True,
                    warn=None) -> np.ndarray:
    """Packs the tensors the engine consumes into one float32 blob (dfx_model_create's input).

    ``strict=False`` follows ``read_cp`` (checkpoint.py:85-103): the reference loads with ``load_state_dict(strict=False)``, DROPS
    every tensor whose size does not match and only warns about missing keys — the affected parameters then keep the values the
    freshly constructed module had.  Here those are the deterministic ones of PyTorch's constructors for BatchNorm (weight 1, bias 0,
    running_mean 0, running_var 1) and zeros for everything else (the reference's random initialisation cannot be reproduced)."""
    import warnings

    warn = warn or (lambda msg: warnings.warn(msg, stacklevel=3))
    total = C.c_int64()
    _lib.check(_lib.lib().dfx_model_blob_floats(C.byref(cfg), C.byref(total)))
    blob = np.zeros(total.value, dtype=np.float32)
    for name, shape, off in tensor_manifest(cfg):
        v = sd.get(name)
        if v is not None:
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != shape:
                if strict:
                    raise ValueError(f"size mismatch for {name}: checkpoint {tuple(v.shape)}, model {shape}")
                warn(f"size mismatch for {name}: copying a param with shape {tuple(v.shape)} from checkpoint, the shape in current "
                     f"model is {shape}. (dropped)")
                v = None
        elif strict:
            raise KeyError(f"state dict is missing '{name}'")
        else:
            warn(f"Missing key: '{name}'")
        n = int(np.prod(shape)) if shape else 1
        if v is None:
            leaf = name.rsplit(".", 1)[-1]
            is_bn = len(shape) == 1 and ".gru." not in name and "lsnr_fc" not in name
            fill = 1.0 if is_bn and leaf in ("weight", "running_var") else 0.0
            blob[off:off + n] = fill
        else:
            blob[off:off + n] = v.astype(np.float32, copy=False).ravel()
    return blob


class DfNet:
    """Inference-only DeepFilterNet3 on libdfx.  Call signature and output shapes of deepfilternet3.py:389-456."""

    def __init__(self, p: ModelParams, state_dict: Dict[str, "np.ndarray | torch.Tensor"], df_state: Optional[DF] = None,
                 run_df: bool = True, strict: bool = True):
        """``run_df=False``: DfNet(run_df=False) of the reference (deepfilternet3.py:383; init_df(mask_only=True)) — mask only.
        ``strict=False``: checkpoint.py:85-103 semantics for incomplete state-dicts (see :func:`pack_state_dict`)."""
        self.p = p
        self.cfg = make_cfg(p)
        blob = pack_state_dict(self.cfg, state_dict, strict=strict)
        h = C.c_void_p()
        _lib.check(_lib.lib().dfx_model_create(C.byref(self.cfg), blob.ctypes.data_as(C.POINTER(C.c_float)), C.byref(h)))
        self._h = h
        self.run_df = bool(run_df)
        if not self.run_df:
            _lib.check(_lib.lib().dfx_model_set_run_df(self._h, 0))
        self.df_state = df_state or DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs)
        # attributes the reference's enhance() probes (enhance.py:234)
        self.nb_df = p.nb_df
        self.df_order = p.df_order
        self.df_lookahead = p.df_lookahead
        self.freq_bins = p.freq_bins
        self.erb_bins = p.nb_erb
        self.post_filter = p.mask_pf
        self._ws: Optional[torch.Tensor] = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        try:
            if h:
                _lib.lib().dfx_model_free(h)
        except Exception:  # noqa: BLE001
            pass

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def set_streams(self, enable: bool) -> None:
        """Run the independent branches of the forward pass on internal HIP streams (default) or serially."""
        _lib.check(_lib.lib().dfx_model_set_streams(self._h, int(bool(enable))))

    def set_pipeline(self, time_chunks: int = 6, min_chunk_frames: int = 32, batch_chunks: int = 1) -> None:
        """Time-chunk / batch-chunk pipelining knobs of the GRU phase (see dfx_model_set_pipeline in include/dfx.h)."""
        _lib.check(_lib.lib().dfx_model_set_pipeline(self._h, int(time_chunks), int(min_chunk_frames), int(batch_chunks)))

    def check(self) -> None:
        """Wait for the device, then raise if a kernel of this model raised a fault (fp16-split range, a flag-wait / spin timeout): the
        results of that pass are invalid (dfx_model_check, include/dfx.h)."""
        _lib.check(_lib.lib().dfx_model_check(self._h))

    def poll(self) -> None:
        """The same without waiting: faults of passes that have completed (dfx_model_poll).  ``enhance()``, ``__call__`` and
        ``DfStream.process`` call it after enqueueing their work, and the C entry points look before they start new work, so a fault
        is raised by the next call on the model at the latest (``DFX_CHECK_EVERY_PASS=1``: by the call that caused it)."""
        _lib.check(_lib.lib().dfx_model_poll(self._h))

    Q_GRU_PERSISTENT, Q_HWQ_PROBE, Q_EXACT_FP32, Q_SPIN_LIMIT, Q_PASSES_PERSISTENT, Q_PASSES_TICKET_BUSY = 1, 2, 3, 4, 5, 6

    def query(self, what: int) -> int:
        """dfx_model_query (include/dfx.h DFX_Q_*)."""
        v = C.c_int64()
        _lib.check(_lib.lib().dfx_model_query(self._h, int(what), C.byref(v)))
        return int(v.value)

    # nn.Module-ish no-ops so that callers written against the reference keep working
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or not _lib.on_device(self._ws):
            self._ws = None   # release the smaller buffer first: the two never have to fit side by side
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=_lib.device())
        return self._ws

    @torch.no_grad()
    def __call__(self, spec: torch.Tensor, feat_erb: torch.Tensor, feat_spec: torch.Tensor, atten_lim: float = 0.0
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """spec [B,1,T,F,2], feat_erb [B,1,T,E], feat_spec [B,1,T,F',2] ->
        (spec_e [B,1,T,F,2], m [B,1,T,E], lsnr [B,T,1], df_coefs [B,O,T,F',2])."""
        p = self.p
        dev = _lib.device()
        B, _, T, F, _ = spec.shape
        spec_d = spec.to(dev, torch.float32).contiguous()
        fe = feat_erb.to(dev, torch.float32).contiguous()
        fs = feat_spec.to(dev, torch.float32).contiguous()
        assert F == p.freq_bins and fe.shape == (B, 1, T, p.nb_erb) and fs.shape == (B, 1, T, p.nb_df, 2)
        spec_e = torch.empty_like(spec_d)
        m = torch.empty((B, 1, T, p.nb_erb), dtype=torch.float32, device=dev)
        lsnr = torch.empty((B, T, 1), dtype=torch.float32, device=dev)
        coefs = torch.empty((B, p.df_order, T, p.nb_df, 2), dtype=torch.float32, device=dev)  # DFX_COEF_BOTF
        nbytes = C.c_int64()
        L = _lib.lib()
        _lib.check(L.dfx_model_workspace_bytes(self._h, B, T, C.byref(nbytes)))
        ws = self.workspace(nbytes.value)
        _lib.check(L.dfx_model_forward(self._h, self.df_state.bands_handle, _lib.ptr(spec_d), _lib.ptr(fe), _lib.ptr(fs),
                                       B, T, float(atten_lim), _lib.ptr(spec_e), _lib.ptr(m), _lib.ptr(lsnr),
                                       _lib.ptr(coefs), _lib.ptr(ws), ws.numel(), _lib.stream()))
        # the engine writes the coefficients directly in DfOutputReshapeMF's layout [B,O,T,F',2] (deepfilternet3.py:268-275)
        if not self.run_df:
            coefs = torch.zeros((), device=dev)   # deepfilternet3.py:444
        self.poll()
        return spec_e, m, lsnr, coefs

    forward = __call__


# ---------------------------------------------------------------------------------------------------- checkpoints
def _find_checkpoint(dirname: str, epoch) -> Tuple[Optional[str], int]:
    """checkpoint.py:46-84: model_<epoch>.ckpt[.best]; 'best' prefers *.best, 'latest' the highest epoch."""
    if not dirname or not os.path.isdir(dirname):
        return None, 0
    cps = glob.glob(os.path.join(dirname, "model_*.ckpt*"))
    if not cps:
        return None, 0

    def ep(path):
        mm = re.search(r"model_(\d+)\.ckpt", os.path.basename(path))
        return int(mm.group(1)) if mm else -1

    if isinstance(epoch, int) or (isinstance(epoch, str) and epoch.isdigit()):
        want = int(epoch)
        for c in cps:
            if ep(c) == want:
                return c, want
        return None, 0
    if epoch == "best":
        best = [c for c in cps if c.endswith(".best")]
        if best:
            c = max(best, key=ep)
            return c, ep(c)
    c = max(cps, key=ep)
    return c, ep(c)


def read_cp(dirname: str, epoch="best") -> Tuple[Optional[Dict[str, torch.Tensor]], int]:
    path, ep = _find_checkpoint(dirname, epoch)
    if path is None:
        return None, 0
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    out = {}
    for k, v in sd.items():  # checkpoint.py:86-92: rename legacy 'clc' keys
        out[k.replace("clc", "df")] = v
    return out, ep


# ---------------------------------------------------------------------------------------------------- .dfx model files
def read_onnx_targz(path: str) -> Tuple[ModelParams, Dict[str, np.ndarray]]:
    """The reference's shipped artefact, ``<model>_onnx.tar.gz`` (tract.rs:29-70 ``DfParams::from_targz``; export.py:331-337), as
    (ModelParams, state-dict with the reference's key names).  All the reading happens in libdfx (``dfx_onnx_targz_read``,
    csrc/dfx_onnx.hip): DSP parameters from config.ini, structure and weights from the three ONNX graphs; BatchNorm layers come back
    as the exporter folded them (identity statistics, the folded bias as ``bias``)."""
    L = _lib.lib()
    cfg = _lib.ModelCfg()
    n = C.c_int64()
    _lib.check(L.dfx_onnx_targz_read(os.fsencode(path), C.byref(cfg), None, 0, C.byref(n)))
    blob = np.empty(n.value, dtype=np.float32)
    _lib.check(L.dfx_onnx_targz_read(os.fsencode(path), C.byref(cfg), blob.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)))
    skip = {0: "none", 1: "identity", 2: "groupedlinear"}
    p = ModelParams(sr=cfg.sr, fft_size=cfg.fft_size, hop_size=cfg.hop_size, nb_erb=cfg.nb_erb, nb_df=cfg.nb_df,
                    min_nb_freqs=cfg.min_nb_freqs, df_order=cfg.df_order, df_lookahead=cfg.df_lookahead, lsnr_min=cfg.lsnr_min,
                    lsnr_max=cfg.lsnr_max, conv_lookahead=cfg.conv_lookahead, conv_ch=cfg.conv_ch, emb_hidden_dim=cfg.emb_hidden_dim,
                    emb_num_layers=cfg.emb_num_layers, df_hidden_dim=cfg.df_hidden_dim, df_num_layers=cfg.df_num_layers,
                    df_gru_skip=skip[cfg.df_gru_skip], df_pathway_kernel_size_t=cfg.df_pathway_kernel_size_t, lin_groups=cfg.lin_groups,
                    enc_lin_groups=cfg.enc_lin_groups, mask_pf=bool(cfg.mask_pf), pf_beta=cfg.pf_beta,
                    emb_gru_skip_enc=skip[cfg.emb_gru_skip_enc], emb_gru_skip=skip[cfg.emb_gru_skip], enc_concat=bool(cfg.enc_concat),
                    norm_alpha_value=float(cfg.norm_alpha))
    sd = {}
    for name, shape, off in tensor_manifest(cfg):
        k = int(np.prod(shape)) if shape else 1
        sd[name] = blob[off:off + k].reshape(shape).copy()
    return p, sd


def export_dfx(path: str, model_base_dir: Optional[str] = None, epoch="best", *, params: Optional[ModelParams] = None,
               state_dict: Optional[Dict[str, "np.ndarray | torch.Tensor"]] = None) -> str:
    """Writes the model file that the C API's ``df_create(path, ...)`` (include/df_capi.h == libDF/src/capi.rs:83-104) and
    ``dfx_model_load_file`` read: configuration + raw float32 state-dict (the role of the reference's ``export.py`` tar.gz of ONNX
    graphs, export.py:331-337 / tract.rs:37-70).  Source: a reference model directory (``config.ini`` + ``checkpoints/``) or
    ``params`` + ``state_dict``.  Needs libdfx.so for the tensor manifest, not a GPU."""
    if params is None:
        if model_base_dir is None:
            raise ValueError("export_dfx needs a model directory or params= + state_dict=")
        params = ModelParams.from_ini(os.path.join(model_base_dir, "config.ini"), must_exist=True)
    if state_dict is None:
        if model_base_dir is None:
            raise ValueError("export_dfx needs a model directory or params= + state_dict=")
        state_dict, _ = read_cp(os.path.join(model_base_dir, "checkpoints"), epoch)
        if state_dict is None:
            raise FileNotFoundError("Could not find a checkpoint")
    cfg = make_cfg(params)
    blob = pack_state_dict(cfg, state_dict)
    _lib.check(_lib.lib().dfx_model_save_file(C.byref(cfg), blob.ctypes.data_as(C.POINTER(C.c_float)), os.fsencode(path)))
    return path
