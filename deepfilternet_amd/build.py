"""Builds deepfilternet_amd/csrc/libdfx.so with hipcc for gfx950 (MI355X).  In-tree, so the .so travels with the repo
snapshot to the GPU box.  hipcc cross-compiles without a GPU; nothing here needs a device."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
REPO = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libdfx.so")
SOURCES = ["dfx_dsp.hip", "dfx_model.hip", "dfx_capi.hip", "dfx_io.hip", "dfx_mf.hip", "dfx_onnx.hip"]
ARCH = "gfx950"
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libdfx.so)")


def sources() -> List[str]:
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps() -> List[str]:
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files if f.endswith((".h", ".hip"))]
    out.append(os.path.join(REPO, "include", "dfx.h"))
    out.append(os.path.join(REPO, "include", "df_capi.h"))
    out.append(os.path.abspath(__file__))   # the compiler flags live here
    return out


STAMP = LIB + ".sources"   # the source list the library was linked from (a library built from another list is stale)


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            if f.read().split() != SOURCES:
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    bdir = os.path.join(CSRC, "_build")
    os.makedirs(bdir, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: matrix-op results in VGPRs instead of AGPRs where registers allow (the epilogues read every accumulator on
    # the VALU: one v_accvgpr_read per value otherwise — 682 of the 4344 instructions of df_convp, 20 % of the persistent GRU kernel's);
    # measured -0.1 ms per step
    # -target-feature -packed-fp32-ops (round 6): no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 anywhere in the library.  On the
    # MI355X of this pool a wave's packed fp32 operations with operand swizzles (op_sel: what complex arithmetic compiles to) return wrong
    # values for groups of 16 lanes while ANOTHER wave on the same SIMD — any kernel, any process — executes the double-rate matrix
    # operations (v_mfma_f32_16x16x32_f16 / _bf16, v_mfma_f32_32x32x16_f16).  That was the "two handles return wrong samples" failure of
    # round 5: one handle's STFT kernels beside the other handle's fp16-split kernels; rocFFT is hit the same way (tools/dev/xkern_probe.hip,
    # docs/measurements.md R6.1).  The scalar forms are as fast here (analysis 0.48 -> 0.45 ms) and tests/test_isa.py keeps the packed ones out.
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
             *NO_PACKED_FP32, f"-I{os.path.join(REPO, 'include')}", f"-I{os.path.join(CSRC, 'env_hip')}", f"-I{CSRC}"]
    if os.environ.get("DFX_BUILD_MFMA_K16") == "1":
        # the "quiet neighbour" build: matrix kernels on v_mfma_f32_16x16x16_f16 (env_hip/dfx_env.h: DFX_MFMA_K16) — does not disturb other
        # kernels' packed fp32 arithmetic on the same GPU, +8.5 % step time (docs/measurements.md R6.1, profiles/r06_k16.log)
        flags.append("-DDFX_MFMA_K16=1")
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [_hipcc(), *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-lz", "-o", LIB + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(" ".join(SOURCES))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
