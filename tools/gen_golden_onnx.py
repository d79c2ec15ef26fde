#!/usr/bin/env python3
"""``<model>_onnx.tar.gz`` fixtures made by the REFERENCE's own model code (build container only; ``/root/reference``).

The reference ships its models as a gzip'ed tar of three ONNX graphs + config.ini (DeepFilterNet/df/scripts/export.py:133-337), which
its real-time runtime opens (libDF/src/tract.rs:29-70).  The released archives are missing blobs here, so this tool produces archives
the way export.py does, from the reference's ``DfNet`` with seeded weights:

  * ``model.enc`` / ``model.erb_dec`` (scripted, ``jit=True``) and ``model.df_dec`` (traced) go through ``torch.onnx.export`` with
    export.py's argument list — input / output names, dynamic axes, ``keep_initializers_as_inputs=False``, opset (export.py:95-106,
    178-284; the script's command-line default is opset 12);
  * config.ini = the model's configuration (export.py:320-324 copies the model directory's), version.txt (export.py:326-329),
    tar layout of export.py:330-337 (members carry the export directory as path prefix).

What cannot be reproduced: ``onnx`` / ``onnxruntime`` / ``onnxsim`` are not installed, so export.py's ``check`` and ``--simplify`` steps
are skipped (both are optional switches of the script) and torch's post-export hook that needs the ``onnx`` package is bypassed (it
only matters for custom onnxscript functions; the serialised graph is complete before it runs).

    python tools/gen_golden_onnx.py            # writes tests/golden/df3s_onnx.tar.gz + tests/golden/onnx_df3s.npz
"""
from __future__ import annotations

import os
import sys
import tarfile
import tempfile
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from tools.gen_golden import GOLDEN, synth_audio  # noqa: E402
from tools.gen_golden_r2 import build  # noqa: E402


def small_params():
    """DeepFilterNet3's structure (lookahead 2/2, grouped-linear DF skip, 5-frame pathway conv) at the smallest size the engine accepts
    with 256-wide GRUs: the fixture is dominated by the three GRU layers (1.2 M weights)."""
    from deepfilternet_amd.config import ModelParams

    p = ModelParams.deepfilternet3()
    p.conv_ch, p.emb_num_layers, p.df_num_layers, p.lin_groups, p.enc_lin_groups = 16, 2, 1, 8, 16
    return p


def quantise_gru(sd):
    """GRU weights on a 2^-10 grid: exact in fp32 (nothing downstream changes) and the archive compresses to about a third."""
    for k in list(sd):
        if ".gru." in k:
            sd[k] = (np.round(np.asarray(sd[k], np.float64) * 1024.0) / 1024.0).astype(np.float32)


def export_targz(model, df_state, p, path, opset=12, name="df3s"):
    """export.py:133-337 without the optional check / simplify steps.  Returns nothing; writes `path`."""
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto  # needs the absent `onnx` package; no custom functions here
    from df.enhance import df_features

    warnings.filterwarnings("ignore")
    audio = torch.randn((1, 1 * p.sr), generator=torch.Generator().manual_seed(1))
    spec, feat_erb, feat_spec = df_features(audio, df_state, p.nb_df, device="cpu")
    feat_spec = feat_spec.transpose(1, 4).squeeze(4)  # export.py:177
    with tempfile.TemporaryDirectory() as d, torch.no_grad():
        export_dir = os.path.join(d, "export")
        os.makedirs(export_dir)

        def export_impl(fn, module, inputs, input_names, output_names, dynamic_axes, jit):
            outputs = module(*inputs)
            m = torch.jit.script(module, example_inputs=[tuple(inputs)]) if jit else module
            torch.onnx.export(model=m, f=os.path.join(export_dir, fn), args=inputs, input_names=input_names, dynamic_axes=dynamic_axes,
                              output_names=output_names, opset_version=opset, keep_initializers_as_inputs=False, dynamo=False)
            return outputs

        e0, e1, e2, e3, emb, c0, lsnr = export_impl(
            "enc.onnx", model.enc, (feat_erb, feat_spec), ["feat_erb", "feat_spec"], ["e0", "e1", "e2", "e3", "emb", "c0", "lsnr"],
            {"feat_erb": {2: "S"}, "feat_spec": {2: "S"}, "e0": {2: "S"}, "e1": {2: "S"}, "e2": {2: "S"}, "e3": {2: "S"}, "emb": {1: "S"},
             "c0": {2: "S"}, "lsnr": {1: "S"}}, True)
        export_impl("erb_dec.onnx", model.erb_dec, (emb.clone(), e3, e2, e1, e0), ["emb", "e3", "e2", "e1", "e0"], ["m"],
                    {"emb": {1: "S"}, "e3": {2: "S"}, "e2": {2: "S"}, "e1": {2: "S"}, "e0": {2: "S"}, "m": {2: "S"}}, True)
        export_impl("df_dec.onnx", model.df_dec, (emb.clone(), c0), ["emb", "c0"], ["coefs"],
                    {"emb": {1: "S"}, "c0": {2: "S"}, "coefs": {1: "S"}}, False)
        with open(os.path.join(export_dir, "config.ini"), "w") as f:
            f.write(p.to_ini())
        with open(os.path.join(export_dir, "version.txt"), "w") as f:
            f.write(f"{name}_epoch_0")
        cwd = os.getcwd()
        os.chdir(d)
        try:
            with tarfile.open(path, mode="w:gz") as f:
                for fn in ("enc.onnx", "erb_dec.onnx", "df_dec.onnx", "config.ini", "version.txt"):
                    f.add(os.path.join("export", fn))
        finally:
            os.chdir(cwd)


def make(p, seed, path, opset=12, quantise=True, name="df3s"):
    model, df_state, sd = build(p, seed, sd_edit=quantise_gru if quantise else None)
    export_targz(model, df_state, p, path, opset=opset, name=name)
    return model, df_state, sd


def main():
    import torch
    from df.enhance import enhance

    p = small_params()
    seed = 11
    path = os.path.join(GOLDEN, "df3s_onnx.tar.gz")
    model, df_state, sd = make(p, seed, path)
    audio = synth_audio(seed + 7, 2, 4800 * 3 + 123)
    with torch.no_grad():
        y = enhance(model, df_state, torch.from_numpy(audio.copy()), pad=True).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "onnx_df3s.npz"), seed=seed, audio=audio, y_pad=y, ini=p.to_ini())
    print(f"{path}: {os.path.getsize(path) / 1e6:.2f} MB; enhance golden written")


if __name__ == "__main__":
    from tools.ref_import import install_shims, reference_available

    if not reference_available():
        sys.exit("reference not available")
    install_shims()
    main()
