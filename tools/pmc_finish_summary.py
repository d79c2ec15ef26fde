"""HBM bytes per launch of dfx_k_synthesis_rows from the two rocprofv3 --pmc passes of tools/dev/pmc_finish.py (FETCH_SIZE, WRITE_SIZE), calibrated
on the pure-stream dispatches of dfx_k_df_apply_rows in the same run (MI355X_MICROARCH.md §HBM: the counters tally requests whose bytes-per-count
depends on the access width; the calibration kernel reads and writes 16-byte-per-lane rows like the finishing kernel reads, the finishing
kernel's audio stores are 16 bytes per lane too).   Usage: python tools/pmc_finish_summary.py <dir with pmc_fetch/ pmc_write/> [out.json]"""
import csv
import glob
import json
import sys

root = sys.argv[1]
B, T, F, E, nd, O = 256, 1002, 481, 32, 96, 5


def series(sub, counter, kernel):
    vals = []
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals = [float(r["Counter_Value"]) for r in rows]
    return vals


cal_f, cal_w = series("pmc_fetch", "FETCH_SIZE", "dfx_k_df_apply"), series("pmc_write", "WRITE_SIZE", "dfx_k_df_apply")
fin_f, fin_w = series("pmc_fetch", "FETCH_SIZE", "dfx_k_synthesis_rows"), series("pmc_write", "WRITE_SIZE", "dfx_k_synthesis_rows")
assert len(cal_f) >= 4 and len(cal_w) >= 4 and fin_f and fin_w, (len(cal_f), len(cal_w), len(fin_f), len(fin_w))
cal_read, cal_write = B * T * (241 + 1) * 16, B * T * 244 * 16
bpu_r, bpu_w = cal_read / (sum(cal_f[:4]) / 4), cal_write / (sum(cal_w[:4]) / 4)
read_bytes, write_bytes = sum(fin_f) / len(fin_f) * bpu_r, sum(fin_w) / len(fin_w) * bpu_w
alg_r, alg_w = (F * 8 + nd * O * 8 + E * 4) * B * T, 480 * 4 * B * T
res = {"kernel": "dfx_k_synthesis_rows<5, false>", "batch": B, "frames_per_clip": T, "launches": len(fin_f),
       "hbm_read_bytes_per_launch": round(read_bytes), "hbm_write_bytes_per_launch": round(write_bytes),
       "hbm_bytes_per_launch": round(read_bytes + write_bytes),
       "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w, "algorithmic_bytes_per_launch": alg_r + alg_w,
       "traffic_over_algorithmic": round((read_bytes + write_bytes) / (alg_r + alg_w), 4),
       "note": "the O - 1 neighbouring frames a frame's taps read (4 x 768 B per frame) are not algorithmic bytes: they are meant to be L2 hits"}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
