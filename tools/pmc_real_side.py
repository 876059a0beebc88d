"""profiles/rNN_pmc_traffic.json from the two PMC passes over tools/run_real_side.py (rocprofv3 --pmc FETCH_SIZE and --pmc
WRITE_SIZE in SEPARATE runs; HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE in KB -- gfx950 tallies 128-byte read
requests as 64 B, WRITE_SIZE is exact: MI355X_MICROARCH.md, HBM section).  Launches are matched by kernel name; the first
launch of each kernel (cold pool rows) is dropped, the rest averaged.
usage: python tools/pmc_real_side.py <fetch_counter_collection.csv> <write_counter_collection.csv> <clips per launch> <out.json>"""
import csv, json, subprocess, sys
from collections import defaultdict

fetch_csv, write_csv, clips, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
KERNELS = (("conv0_fwd_f16", "void conv0_breg"), ("conv1_fwd_f16", "void conv_mfma_kernel<1, 3, false, 2, 1"),
           ("conv2_fwd_f16x3_real", "void conv_mfma_kernel<3, 4, false, 1, 0"), ("conv2_fwd_f16c8_real", "void conv_mfma_kernel<4, 4, false, 1, 0"))
# algorithmic bytes per clip: level 0 reads 16-bit pixel rows once and writes pooled f16 slots (+ nothing else);
# level 1 reads them and writes both planes of its pooled output; level 2 reads both planes and writes fp32 features
ALGO = {"conv0_fwd_f16": 16 * 3 * 112 * 120 * 2 + 64 * 16 * 28 * 28 * 2, "conv1_fwd_f16": 64 * 16 * 28 * 28 * 2 + 2 * 128 * 8 * 7 * 7 * 2,
        "conv2_fwd_f16x3_real": 2 * 128 * 8 * 7 * 7 * 2 + 2048 * 4, "conv2_fwd_f16c8_real": 2 * 128 * 8 * 7 * 7 * 2 + 2048 * 4}


def per_kernel(path, counter):
    acc = defaultdict(list)
    for row in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"])):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    return acc


f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
try:
    rev = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    rev = None
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_distillation_amd import hip
res = {"_source": {"command": "tools/run_real_side.py (the launches of one bench.py step, config 2)", "git": rev,
                   "kernel_sources_sha256_16": hip.sources_hash(),      # bench.py quotes the file only while this matches its own sources
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB"}}
for label, prefix in KERNELS:
    fk = [k for k in f if k.startswith(prefix)]
    if not fk:
        continue
    k = fk[0]
    fv = [v for _, v in f[k]][1:] or [v for _, v in f[k]]
    wv = [v for _, v in w.get(k, [])][1:] or [v for _, v in w.get(k, [(0, 0.0)])]
    fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
    hbm = (2.0 * fm + wm) * 1024.0
    res[label] = {"kernel": k.split("(")[0], "grid": f[k][0][0], "clips_per_launch": clips, "launches_averaged": len(fv),
                  "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm, "hbm_bytes_per_launch": hbm,
                  "algorithmic_bytes_per_launch": ALGO[label] * clips, "over_algorithmic": hbm / (ALGO[label] * clips)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
