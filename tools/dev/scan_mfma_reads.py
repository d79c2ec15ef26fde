#!/usr/bin/env python3
"""Dev: static check of the gfx950 assembly for reads of a matrix-op result that come too early.

A VALU / LDS / memory instruction that reads the destination registers of a v_mfma must be >= NEED wait states behind it (8-pass ops: 11).
The compiler's hazard recognizer inserts the s_nops — but it was seen (round 5, dfx_fft480_mfma) to undercount when the first reader sits in
the NEXT basic block (`v_mfma ...; s_and_saveexec; <block>: s_nop 3; v_sub reads the result` = 8 states): the transform then returned a wrong
bin in ~10 % of the runs on the GPU, never on the interpreter.  This scan walks every kernel linearly (fall-through across labels and
branches) and reports the early reads that sit behind a change of the exec mask — the pattern that failed (--all: every read closer than
NEED; the compiler's own rule for the four-pass v_mfma_f32_16x16x32_f16 is 7, and the ~750 same-block reads at 8-10 states in this library
have never been seen to fail).

    python tools/dev/scan_mfma_reads.py [dfx_dsp.hip dfx_model.hip ...]        exit status 1 if anything is found
"""
import os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "deepfilternet_amd", "csrc")
NEED = 11
ALL = "--all" in sys.argv   # every read closer than NEED, not only the ones behind a change of the exec mask (the pattern that failed)


def assembly(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", f"-I{REPO}/include",
               f"-I{CSRC}/env_hip", f"-I{CSRC}", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def regs(tok):
    """VGPR / AGPR numbers named by one operand token."""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def scan(lines):
    found, name, pending = [], None, []   # pending: [dst regs, age, text, exec changed since]
    for ln, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, pending = m.group(1), []
            continue
        t = l.strip()
        if name is None or not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        cost = 1
        if op == "s_nop":
            cost = int(ops[0]) + 1
        if op == "s_endpgm":
            pending = []
            continue
        is_mfma = op.startswith("v_mfma")
        if not is_mfma and op != "s_nop":
            reads = set()
            srcs = ops if op.startswith(("ds_write", "global_store", "buffer_store", "scratch_store", "global_atomic")) else ops[1:]
            for o in srcs:
                reads |= regs(o)
            for p in pending:
                if p[1] < NEED and reads & p[0] and (p[3] or ALL):
                    found.append((name, ln + 1, p[1], p[2], t))
        if "exec" in t and op.startswith("s_"):
            for p in pending:
                p[3] = True
        for p in pending:
            p[1] += cost
        pending = [p for p in pending if p[1] < NEED]
        if is_mfma:
            dst = regs(ops[0])
            # a later op that overwrites the registers ends the interest in the older one
            pending = [p for p in pending if not (p[0] & dst)]
            pending.append([dst, 0, t, False])
    return found


def main():
    srcs = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for s in srcs:
        for name, ln, age, mf, rd in scan(assembly(s)):
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:90]
            print(f"{s}:{ln} {d}\n    {age} wait states after  {mf}\n    read by             {rd}")
            bad += 1
    print(f"{bad} early read(s) of a matrix-op result")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
