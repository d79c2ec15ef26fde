#!/usr/bin/env python3
"""Check this engine against the reference's PRETRAINED DeepFilterNet3 — the day the checkpoint blobs exist.

    python tools/check_pretrained.py <DeepFilterNet3.zip | DeepFilterNet3_onnx.tar.gz | model directory> [--assets DIR] [--no-reference]

The reference pins its released models with `DeepFilterNet/df/scripts/test_df.py:44-78`: enhance `assets/noisy_snr0.wav`, compare
against `assets/clean_freesound_33711.wav`, and expect (atol = rtol = 1e-4, test_df.py:20-21) for DeepFilterNet3
    SI-SDR 20.014915466308594 dB   (evaluation_utils.py:599-619, si_sdr_speechmetrics)
    STOI   0.9742409586906433      (pystoi: not installed here, not computed)
In this build container the checkpoints are missing large blobs (`/root/reference/.MISSING_LARGE_BLOBS`: models/DeepFilterNet3.zip,
models/DeepFilterNet3_onnx.tar.gz, ...), so parity for the *pretrained* weights is still unpinned.  This script is what
closes that gap when they are there:

  1. unpack (`DeepFilterNet3.zip` = `<name>/config.ini` + `<name>/checkpoints/model_<epoch>.ckpt.best`, enhance.py:146-176;
     an `_onnx.tar.gz` is read as it is, tract.rs:29-70) and build the model with `deepfilternet_amd.init_df`;
  2. enhance noisy_snr0.wav on the GPU (or the CPU interpreter build with DFX_BACKEND=emu: slow), print SI-SDR next to the pin;
  3. where `/root/reference` can be imported (a model directory / zip only: the reference's Python loads checkpoints, not ONNX archives):
     run the reference's own `df.enhance.enhance` with the same checkpoint on its CPU path (libdf = the C oracle, tools/ref_import.py)
     and print the RMS difference of the two waveforms (bar: 1e-4, BASELINE.json north_star).

Exit status: 0 all checks that could run passed, 1 a check failed, 2 nothing could be checked (model or assets missing).
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import zipfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

PINS = {   # df/scripts/test_df.py:44-78
    "DeepFilterNet3": {"sdr": 20.014915466308594, "stoi": 0.9742409586906433},
    "DeepFilterNet2": {"sdr": 19.41733717918396, "stoi": 0.9725977621169399},
    "DeepFilterNet": {"sdr": 18.88543128967285, "stoi": 0.9689496585281197},
}
A_TOL = R_TOL = 1e-4   # test_df.py:20-21


def si_sdr(reference: np.ndarray, estimate: np.ndarray) -> float:
    """evaluation_utils.py:599-619 (si_sdr_speechmetrics), one reference and one estimate."""
    reference = reference.reshape(-1, 1)
    estimate = estimate.reshape(-1, 1)
    eps = np.finfo(reference.dtype).eps
    rss = np.dot(reference.T, reference)
    a = (eps + np.dot(reference.T, estimate)) / (rss + eps)
    e_true = a * reference
    e_res = estimate - e_true
    sss = (e_true ** 2).sum()
    snn = (e_res ** 2).sum()
    return float(10 * np.log10((eps + sss) / (eps + snn)))


def unpack(path: str, tmp: str) -> str:
    """-> what init_df takes: a model directory (config.ini + checkpoints/) or the path of an _onnx.tar.gz."""
    if os.path.isdir(path):
        return path
    if path.endswith(".tar.gz"):
        return path
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            z.extractall(tmp)
        for root, _dirs, files in os.walk(tmp):
            if "config.ini" in files and os.path.isdir(os.path.join(root, "checkpoints")):
                return root
        raise SystemExit(f"check_pretrained: no config.ini + checkpoints/ inside {path}")
    raise SystemExit(f"check_pretrained: {path} is neither a model directory, a zip nor an _onnx.tar.gz "
                     "(a file of a few hundred bytes here is a git-lfs pointer: the blob is missing)")


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("model")
    ap.add_argument("--assets", default=None, help="directory with noisy_snr0.wav and clean_freesound_33711.wav (default: /root/reference/assets, ./assets)")
    ap.add_argument("--no-reference", action="store_true", help="skip the run of the reference's own enhance()")
    ap.add_argument("--name", default=None, help="which pin of test_df.py applies (default: from the file name)")
    args = ap.parse_args()

    if not os.path.exists(args.model) or (os.path.isfile(args.model) and os.path.getsize(args.model) < 4096):
        print(f"check_pretrained: {args.model}: missing or a pointer stub (see .MISSING_LARGE_BLOBS) — nothing to check")
        return 2
    assets = next((d for d in (args.assets, "/root/reference/assets", os.path.join(REPO, "assets")) if d and os.path.isfile(os.path.join(d, "noisy_snr0.wav"))), None)
    if assets is None:
        print("check_pretrained: assets/noisy_snr0.wav not found (--assets)")
        return 2
    name = args.name or next((n for n in sorted(PINS, key=len, reverse=True) if n.lower() in os.path.basename(os.path.abspath(args.model)).lower()), "DeepFilterNet3")

    import torch

    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.io import load_audio

    ok = True
    with tempfile.TemporaryDirectory() as tmp:
        src = unpack(args.model, tmp)
        model, df_state, suffix, epoch = init_df(src, config_allow_defaults=True)
        sr = df_state.sr()
        noisy, _ = load_audio(os.path.join(assets, "noisy_snr0.wav"), sr)
        clean_path = os.path.join(assets, "clean_freesound_33711.wav")
        print(f"model {suffix} (epoch {epoch}), {noisy.shape[-1] / sr:.2f} s of audio at {sr} Hz")
        enhanced = enhance(model, df_state, noisy, pad=True)
        if enhanced.is_cuda:
            torch.cuda.synchronize()
        enhanced = enhanced.cpu()
        if os.path.isfile(clean_path):
            clean, _ = load_audio(clean_path, sr)
            n = min(clean.shape[-1], enhanced.shape[-1])
            got = si_sdr(clean[0, :n].numpy(), enhanced[0, :n].numpy())
            pin = PINS.get(name, {}).get("sdr")
            if pin is None:
                print(f"SI-SDR {got:.6f} dB (no pin for {name})")
            else:
                close = abs(got - pin) <= A_TOL + R_TOL * abs(pin)
                ok = ok and close
                print(f"SI-SDR {got:.6f} dB, test_df.py pins {pin:.6f} for {name}: {'OK' if close else 'DIFFERENT'} (atol = rtol = 1e-4)")
        else:
            print(f"{clean_path} not found: SI-SDR not computed")
        # ---- the reference's own enhance() with the same checkpoint
        ref_ok = not args.no_reference and os.path.isdir(src)
        if ref_ok:
            try:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import ref_import

                if not ref_import.reference_available():
                    raise ImportError("/root/reference not present")
                ref_import.install_shims()
                from df.enhance import enhance as ref_enhance  # type: ignore
                from df.enhance import init_df as ref_init_df  # type: ignore

                rmodel, rstate, _, _ = ref_init_df(src, log_file=None, config_allow_defaults=True)
                ref = ref_enhance(rmodel, rstate, noisy.cpu(), pad=True)
                d = float((ref - enhanced).pow(2).mean().sqrt())
                close = d <= 1e-4
                ok = ok and close
                print(f"RMS difference to the reference's enhance() on its CPU path (libdf = oracle/): {d:.3e}: {'OK' if close else 'DIFFERENT'} (bar 1e-4)")
            except Exception as e:  # noqa: BLE001
                print(f"reference run skipped: {e!r}")
        elif not args.no_reference:
            print("reference run skipped: the reference's Python loads model directories / zips, not ONNX archives")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
