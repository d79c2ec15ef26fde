#!/usr/bin/env python3
"""Kernel timeline of ONE streaming call from the tail of a rocprofv3 kernel trace (tools/gpu_trace_stream.sh keeps the last 400 rows):
the launches between the last-but-two and the last-but-one STFT kernel.  Usage: tools/stream_call_timeline.py <stream_trace_tail.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "dfx_k_analysis<" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
print("# start_us end_us dur_us queue kernel   (one call: from one STFT launch to the next)")
for r in rows[a:b + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>2} {r['Kernel_Name'][:90]}")
