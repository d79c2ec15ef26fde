"""Dev: per-stream busy time and kernel totals of one step of tools/timeline.py output."""
import collections
import sys

rows = []
for line in open(sys.argv[1]):
    if line.startswith("#") or not line.strip():
        continue
    parts = line.split()
    a, b, d = float(parts[0]), float(parts[1]), float(parts[2])
    name = " ".join(parts[3:-1])
    rows.append((a, b, d, name, parts[-1]))
if not rows:
    sys.exit("empty timeline")
span = max(r[1] for r in rows) - min(r[0] for r in rows)
print(f"step span {span:.3f} ms, {len(rows)} launches")
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r[4]].append(r)
for sname, rs in sorted(by_stream.items(), key=lambda kv: int(kv[0][1:])):
    busy = sum(r[2] for r in rs)
    waits = sum(r[2] for r in rs if "wait_ge" in r[3])
    first, last = min(r[0] for r in rs), max(r[1] for r in rs)
    kinds = collections.Counter(r[3].split("(")[0].replace("void ", "")[:22] for r in rs)
    print(f"  {sname:4s} {first:7.3f} .. {last:7.3f}  busy {busy:7.3f} ms (wait kernels {waits:6.3f})  {len(rs):4d} launches  {dict(kinds.most_common(4))}")
tot = collections.defaultdict(float)
for r in rows:
    tot[r[3].split("(")[0].replace("void ", "")[:30]] += r[2]
print("kernel totals (ms):", {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:16]})
