#!/usr/bin/env python3
"""Dev: every kernel of the library whose gfx950 code object uses scratch (register spills), from the compiler's resource-usage remarks.
    python tools/dev/scratch_report.py            # prints `source bytes/lane VGPRs kernel`; exit status 1 if any kernel spills"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "deepfilternet_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",   # (the product flags: deepfilternet_amd/build.py)
         "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(CSRC, "env_hip"), "-I" + CSRC, "-Rpass-analysis=kernel-resource-usage",
         "--cuda-device-only", "-c", "-o", "/dev/null"]


def remarks(src):
    r = subprocess.run(["hipcc"] + FLAGS + [os.path.join(CSRC, src)], capture_output=True, text=True)
    return src, r.returncode, r.stderr


def main():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    with ThreadPoolExecutor(len(srcs)) as ex:
        for src, rc, err in ex.map(remarks, srcs):
            name, vg = None, "?"
            if rc != 0:   # a source that does not compile has no remarks: that is a failure, not "no spills"
                print(f"{src}: hipcc failed (rc {rc})\n{err[-2000:]}")
                bad += 1
                continue
            with open(os.path.join(CSRC, src)) as f:
                text = f.read()
            defines_kernels = "__global__" in text or any(
                "__global__" in open(os.path.join(CSRC, h)).read() for h in re.findall(r'#include "(dfx_\w+\.h)"', text) if os.path.exists(os.path.join(CSRC, h)))
            if defines_kernels and "Function Name:" not in err:
                print(f"{src}: no kernel-resource-usage remarks although the source defines kernels (toolchain without the remark pass?)")
                bad += 1
                continue
            for line in err.splitlines():
                m = re.search(r"Function Name: (\S+)", line)
                if m:
                    name = m.group(1)
                m = re.search(r" VGPRs: (\d+)", line)
                if m:
                    vg = m.group(1)
                m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
                if m and int(m.group(1)) > 0:
                    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                    print(f"{src} {m.group(1)} {vg} {d[:160]}")
                    bad += 1
    print(f"{bad} kernel(s) with scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
