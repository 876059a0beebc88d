"""Summarise the four SQ / TA counter passes of tools/pmc_sq.sh (gpurun_out/pmc_sq/pass{1..4}.csv) per kernel: mean counter
value per dispatch, kernel duration per pass, and the derived figures DESIGN.md quotes (matrix-pipe busy fraction and the
clock the chip held: SQ_BUSY_CU_CYCLES / CUs / duration).  usage: python tools/pmc_sq_summary.py DIR OUT.json"""
import csv, json, sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
res = defaultdict(dict)
for i in range(1, 5):
    rows = list(csv.DictReader(open("%s/pass%d.csv" % (d, i))))
    acc, dur = defaultdict(list), defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("void conv"):
            continue
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for (k, c), v in acc.items():
        res[k][c] = sum(v) / len(v)
    for k, v in dur.items():
        res[k]["dur_ns_pass%d" % i] = sum(v) / len(v)
for k, v in res.items():
    if "SQ_BUSY_CU_CYCLES" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        # SQ_BUSY_CU_CYCLES sums over CUs; SQ_VALU_MFMA_BUSY_CYCLES over SIMDs (4 per CU)
        v["mfma_pipe_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_BUSY_CU_CYCLES"])
        v["sustained_clock_GHz"] = v["SQ_BUSY_CU_CYCLES"] / 256.0 / v["dur_ns_pass1"]
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, v in res.items():
    print("%-45s mfma busy %.2f  clock %.2f GHz  lds conflicts %.2f  dur %.3f ms" % (
        k, v.get("mfma_pipe_busy_frac", 0), v.get("sustained_clock_GHz", 0), v.get("lds_conflict_frac", 0), v.get("dur_ns_pass1", 0) / 1e6))
