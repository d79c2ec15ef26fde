#!/usr/bin/env python3
"""Flags staging loops that serialise their memory latency: a short loop body that contains a global load, an `s_waitcnt vmcnt(0)`
and an LDS store means the wave waits out one full memory latency per iteration; loops with many loads per trip
(already batched) and the partial-tile fallbacks of a kernel also match the pattern: read the hits, do not count them.  Compiles the kernel sources to
gfx950 assembly (no GPU needed) and scans every loop.

    python tools/dev/scan_isa.py [dfx_dsp.hip dfx_model.hip dfx_io.hip dfx_mf.hip]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "deepfilternet_amd", "csrc")


def scan(src: str):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO}/include", f"-I{CSRC}/env_hip", f"-I{CSRC}",
               "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    kern, labels, hits = None, {}, []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, labels = m.group(1), {}
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels:
            body = lines[labels[m.group(1)]:i]
            if len(body) < 80:
                gl = sum("global_load" in b or "buffer_load" in b for b in body)
                ds = sum("ds_write" in b for b in body)
                w0 = sum("vmcnt(0)" in b for b in body)
                if gl and gl <= 2 and ds and w0:   # 1-2 loads per trip: nothing else in flight while the wave waits
                    hits.append((kern, m.group(1), len(body), gl, ds))
    return hits


if __name__ == "__main__":
    for src in (sys.argv[1:] or ["dfx_dsp.hip", "dfx_model.hip", "dfx_io.hip", "dfx_mf.hip"]):
        for kern, label, n, gl, ds in scan(src):
            print(f"{src}: {kern} loop {label}: {n} instructions, {gl} global load(s), {ds} LDS store(s), waits vmcnt(0) inside")
