"""Builds tests/hipemu/_build/libdfx_emu.so: the SAME kernel + launcher sources as libdfx.so, compiled with g++ against
the fiber-based SIMT interpreter in tests/hipemu/dfx_env.h.  Test infrastructure only — the package never loads it."""
from __future__ import annotations

import fcntl
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "deepfilternet_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libdfx_emu.so")
SOURCES = ["dfx_dsp.hip", "dfx_model.hip", "dfx_capi.hip", "dfx_io.hip", "dfx_mf.hip", "dfx_onnx.hip"]


def _deps():
    out = [os.path.join(HERE, "dfx_env.h"), os.path.join(REPO, "include", "dfx.h"), os.path.join(REPO, "include", "df_capi.h")]
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hip")):
            out.append(os.path.join(CSRC, f))
    return out


def build(force: bool = False) -> str:
    # one builder at a time (pytest-xdist workers import this side by side; two of them compiling into the same object files, or one renaming the
    # library the other has just linked, failed a worker now and then): whoever comes second finds the library up to date
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(force)


def _build_locked(force: bool) -> str:
    stamp = OUT + ".sources"
    try:
        with open(stamp) as f:
            same = f.read().split() == SOURCES
    except OSError:
        same = False
    if not force and same and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in _deps()):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    flags = ["-O2", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mf16c", "-Wall", "-Wno-unused-function",
             "-Wno-unknown-pragmas", "-Wno-attributes", f"-I{os.path.join(REPO, 'include')}", f"-I{HERE}", f"-I{CSRC}"]
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(HERE, "_build", s + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen(["g++", *flags, "-x", "c++", "-c", src, "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"g++ failed on {src}:\n{out}")
    r = subprocess.run(["g++", "-shared", "-fPIC", *objs, "-lz", "-o", OUT + ".tmp"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(OUT + ".tmp", OUT)
    with open(stamp, "w") as f:
        f.write(" ".join(SOURCES))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
