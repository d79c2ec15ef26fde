"""Turns the two rocprofv3 --pmc passes of tools/dev/pmc_df_apply.py (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch of the
config-2 DF-apply, calibrated on the pure-stream dispatches of the same kernel whose byte counts are known exactly
(MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are request tallies whose bytes-per-count depends on the access width).
Usage: python tools/pmc_df_apply_summary.py <dir with pmc_fetch/ pmc_write/> [out.json]"""
import csv
import glob
import json
import sys

root = sys.argv[1]
B, T, F, E, nd, O = 256, 1002, 481, 32, 96, 5


def series(sub, counter):
    vals = []
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "dfx_k_df_apply" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals = [float(r["Counter_Value"]) for r in rows]
    return vals


fetch, write = series("pmc_fetch", "FETCH_SIZE"), series("pmc_write", "WRITE_SIZE")
assert len(fetch) >= 8 and len(write) >= 8, (len(fetch), len(write))
# the calibration dispatches of dfx_k_df_apply_rows read 241 float4 of every 488-bin row (bins 0..481; the one coefficient float4 per frame
# of the 2-bin deep filter is counted too) and write 244 float4 (the row's 61 sectors of 64 bytes)
cal_read = B * T * (241 + 1) * 16
cal_write = B * T * 244 * 16
f_cal, w_cal = sum(fetch[:4]) / 4, sum(write[:4]) / 4
f_dfa, w_dfa = sum(fetch[4:8]) / 4, sum(write[4:8]) / 4
read_bytes = f_dfa * cal_read / f_cal
write_bytes = w_dfa * cal_write / w_cal
alg = (F * 8 + nd * O * 8 + E * 4 + F * 8) * B * T
res = {"kernel": "dfx_k_df_apply_rows", "model": "df3", "batch": B, "frames_per_clip": T,
       "counters": {"FETCH_SIZE_calibration": f_cal, "WRITE_SIZE_calibration": w_cal, "FETCH_SIZE": f_dfa, "WRITE_SIZE": w_dfa},
       "calibration": "bytes per counter unit from 4 pure-stream dispatches of the same kernel (nb_df=2, order=1, no gains) with known byte counts",
       "hbm_read_bytes_per_launch": round(read_bytes), "hbm_write_bytes_per_launch": round(write_bytes),
       "hbm_bytes_per_launch": round(read_bytes + write_bytes), "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": round((read_bytes + write_bytes) / alg, 4)}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
