#!/usr/bin/env python3
"""Static checks on the gfx950 assembly of the kernel sources (no GPU needed: hipcc cross-compiles).  Each check is a pattern that cost
measurable time somewhere in this code base; the hits are to be READ, not counted — partial-tile fallbacks and
already-batched loops match too.

  staging       a short loop with 1-2 global loads, an `s_waitcnt vmcnt(0)` and an LDS store per trip: the wave waits out one full memory
                latency per iteration (fix: a compile-time inner trip count, all loads of a pass before the first store)
  lds-chain     straight-line code in which LDS reads keep following LDS stores, each read waited for on its own (`ds_write … ds_read …
                s_waitcnt lgkmcnt(0)`, three times or more in a row): tables and data in one LDS allocation may alias as far as the
                compiler knows, so a read written after a store is not moved above it and becomes a round trip of its own (fix: read
                everything a phase needs before its first store)
  predicated    `s_and_saveexec` … one load … `s_or exec`: a load under `cond ? p[i] : 0` costs five instructions where a clamped index
                costs one (fix: clamp the index, drop the value)
  div64         kernels with many `v_mul_hi_u32` (a 64-bit integer division is ~130 VALU instructions: 30 of them `v_mul_hi_u32`-class);
                look for row maps / flat-index decompositions inside loops (fix: 32-bit operands, or once per row)

    python tools/dev/scan_isa.py [-c staging,lds-chain,predicated,div64] [dfx_dsp.hip dfx_model.hip dfx_io.hip dfx_mf.hip]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "deepfilternet_amd", "csrc")
CHECKS = ("staging", "lds-chain", "predicated", "div64")


def assembly(src: str):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", f"-I{REPO}/include",
               f"-I{CSRC}/env_hip", f"-I{CSRC}", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def kernels(lines):
    """(name, [instruction lines]) per kernel; labels are kept (they end straight-line regions), comments and directives dropped."""
    name, body = None, []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = l.strip()
        if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        body.append(t)
        if t.startswith("s_endpgm"):
            yield name, body
            name = None


def scan_staging(body):
    labels = {}
    for i, t in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels:
            loop = body[labels[m.group(1)]:i]
            if len(loop) < 80:
                gl = sum(b.startswith(("global_load", "buffer_load")) for b in loop)
                ds = sum(b.startswith("ds_write") for b in loop)
                if gl and gl <= 2 and ds and any("vmcnt(0)" in b for b in loop):
                    yield f"loop {m.group(1)}: {len(loop)} instructions, {gl} global load(s), {ds} LDS store(s), waits vmcnt(0) inside"


def scan_lds_chain(body):
    run, start, state = 0, 0, "idle"   # idle -> stored -> read -> (wait) -> counted
    for i, t in enumerate(body + [".LBB_end:"]):
        if t.startswith(".LBB") or t.startswith(("s_barrier", "s_cbranch", "s_branch")):
            if run >= 3:
                yield f"instructions {start}..{i}: {run} LDS reads in a row each issued after an LDS store and waited for with lgkmcnt(0)"
            run, state = 0, "idle"
        elif t.startswith("ds_write"):
            state = "stored"
        elif t.startswith("ds_read") and state == "stored":
            state = "read"
        elif t.startswith("s_waitcnt") and "lgkmcnt(0)" in t and state == "read":
            if run == 0:
                start = i
            run, state = run + 1, "idle"


def scan_predicated(body):
    n = 0
    for i, t in enumerate(body):
        if t.startswith("s_and_saveexec") and i + 3 < len(body):
            nxt = [b for b in body[i + 1:i + 4]]
            loads = [b for b in nxt if b.startswith(("ds_read", "global_load", "buffer_load"))]
            if len(loads) == 1 and any(b.startswith("s_or_b64 exec") for b in nxt):
                n += 1
    if n >= 8:
        yield f"{n} loads each under its own exec mask"


def scan_div64(body):
    n = sum(t.startswith("v_mul_hi_u32") for t in body)
    if n >= 24:
        yield f"{n} v_mul_hi_u32 ({sum(t.startswith('v_') for t in body)} VALU instructions in the kernel): integer divisions on the vector unit"


SCANNERS = {"staging": scan_staging, "lds-chain": scan_lds_chain, "predicated": scan_predicated, "div64": scan_div64}

if __name__ == "__main__":
    args = sys.argv[1:]
    checks = CHECKS
    if args[:1] == ["-c"]:
        checks, args = tuple(args[1].split(",")), args[2:]
    for src in (args or ["dfx_dsp.hip", "dfx_model.hip", "dfx_io.hip", "dfx_mf.hip"]):
        for name, body in kernels(assembly(src)):
            for c in checks:
                for msg in SCANNERS[c](body):
                    print(f"{src}: {name[:72]} [{c}] {msg}")
