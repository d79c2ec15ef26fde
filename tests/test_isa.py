"""What the shipped code objects may not contain (CPU test: the library is disassembled, nothing runs).

Packed fp32 operations (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32): on the MI355X of this pool a wave's packed operations with
operand swizzles return wrong values for 16-lane groups while another wave of the SIMD — any kernel of any process — executes double-rate
matrix operations (v_mfma_f32_16x16x32_f16 and relatives).  That was round 5's "two handles return wrong samples" (docs/measurements.md R6.1,
tools/dev/xkern_probe.hip); the library is built with -target-feature -packed-fp32-ops (deepfilternet_amd/build.py) and this test keeps it so.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from deepfilternet_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _code_objects(lib, tmp):
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, capture_output=True, check=False)   # writes lib.so.<i>.hipv4-amdgcn-...-gfx950
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f and f.endswith("gfx950"))


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_no_packed_fp32_operations_in_the_library():
    lib = _lib.DEFAULT_LIB
    if not os.path.exists(lib):
        pytest.skip("csrc/libdfx.so not built")
    with tempfile.TemporaryDirectory() as tmp:
        objs = _code_objects(lib, tmp)
        assert objs, "no gfx950 code object found in libdfx.so"
        packed, kernels, mfma = {}, 0, 0
        for o in objs:
            cur = None
            for line in subprocess.run([OBJDUMP, "-d", o], capture_output=True, text=True, check=True).stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    kernels += 1
                    continue
                if re.search(r"\bv_pk_(fma|mul|add)_f32\b|\bv_pk_mov_b32\b", line):
                    packed[cur] = packed.get(cur, 0) + 1
                if "v_mfma_" in line:
                    mfma += 1
        assert kernels > 50 and mfma > 1000, (kernels, mfma)   # the disassembly is the library's (sanity)
        assert not packed, f"packed fp32 operations in {len(packed)} kernels, e.g. {sorted(packed.items(), key=lambda kv: -kv[1])[:3]}"
