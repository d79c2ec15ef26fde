#!/usr/bin/env python3
"""Print one enhance() step of a rocprofv3 --kernel-trace CSV as a timeline (start/end/duration per kernel, stream id).

    python tools/timeline.py <..._kernel_trace.csv> [step_index_from_end=1]

A step is delimited by dfx_k_copy_rows/dfx_k_analysis launches (the first kernels of dfx_enhance)."""
import csv
import sys


def main() -> None:
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("kernel_name") or ""
            if not name.startswith(("dfx_", "void dfx_")):
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id") or r.get("Queue_Id") or "?"))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "dfx_k_analysis" in r[2]]
    if not starts:
        raise SystemExit("no dfx_k_analysis launch in the trace")
    i0 = starts[-back]
    while i0 > 0 and "copy_rows" in rows[i0 - 1][2] and rows[i0][0] - rows[i0 - 1][0] < 2_000_000:
        i0 -= 1
    i1 = len(rows)
    for s in starts:
        if s > starts[-back]:
            i1 = s
            break
    while i1 > i0 and "copy_rows" in rows[i1 - 1][2] and i1 < len(rows):
        i1 -= 1
    sel = rows[i0:i1]
    t0 = sel[0][0]
    sid = {}
    print("# start_ms end_ms dur_ms kernel stream")
    for a, b, n, q in sel:
        sid.setdefault(q, len(sid))
        print(f"{(a - t0) / 1e6:8.3f} {(b - t0) / 1e6:8.3f} {(b - a) / 1e6:7.3f} {n[:48]:48s} s{sid[q]}")
    print(f"# step span {(max(r[1] for r in sel) - t0) / 1e6:.3f} ms, {len(sel)} launches")


if __name__ == "__main__":
    main()
