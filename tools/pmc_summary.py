"""Derive profiles/r01_pmc_traffic.json from the two rocprofv3 PMC passes over tools/run_l1.py
(separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, MI355X_MICROARCH.md HBM section):
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB -- gfx950 tallies 128-byte read requests as 64 B.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <clips> <out.json>"""
import csv, json, sys
from collections import defaultdict

fetch_csv, write_csv, clips, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def per_kernel(path, counter):
    """(kernel name, grid) -> values, plus the keys in order of first dispatch."""
    acc, order = defaultdict(list), []
    for row in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"])):
        if row["Counter_Name"] != counter:
            continue
        key = (row["Kernel_Name"], int(row["Grid_Size"]))
        if key not in acc:
            order.append(key)
        acc[key].append(float(row["Counter_Value"]))
    return acc, order


(f, order), (w, _) = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
res = {}
# run_l1.py launches, in dispatch order: pix2rows, then the forward programs of layers 0, 1, 2
keys = [k for k in order if k[0].startswith("pix2rows_kernel")][:1] + [k for k in order if k[0].startswith("void conv")][:3]
import os
labels = ["pix2rows_f16", "conv0_fwd_f16", "conv1_fwd_f16", "conv2_fwd_f16x3_real" if os.environ.get("VD_RUN_L1_HILO") == "1" else "conv2_fwd_f16"]
for key, label in zip(keys, labels):
    fv = sum(f[key]) / len(f[key])
    wv = sum(w[key]) / len(w[key]) if key in w else 0.0
    res[label] = {"kernel": key[0].split("(")[0], "grid": key[1], "clips_per_launch": clips, "FETCH_SIZE_KB": fv, "WRITE_SIZE_KB": wv,
                  "hbm_bytes_per_launch": (2.0 * fv + wv) * 1024.0,
                  "note": "separate --pmc passes; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md "
                          "HBM section); WRITE_SIZE exact"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
