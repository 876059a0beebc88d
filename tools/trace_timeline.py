"""Per-stream timeline of one steady-state step out of a rocprofv3 --kernel-trace csv (tools/trace_step.sh):
   python tools/trace_timeline.py <kernel_trace.csv> [step_from_end]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([\w:]+)(<[^>]{0,40})?", n)
    return (m.group(1) + (m.group(2) or ""))[:70] if m else n[:70]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step = from one launch of the first-level forward of the real side (largest grid conv kernel on its stream) to the next
l0 = [r for r in rows if "conv_mfma_kernel" in r["Kernel_Name"]]
by_stream = {}
for r in l0:
    by_stream.setdefault(r["Stream_Id"], []).append(r)
real = max(by_stream, key=lambda s: sum(x["e"] - x["s"] for x in by_stream[s]))
big = max(int(r["Grid_Size_X"]) for r in by_stream[real])
starts = [r["s"] for r in by_stream[real] if int(r["Grid_Size_X"]) == big]
t0, t1 = starts[-back - 1], starts[-back]
print(f"real-clip stream {real}; step window {(t1 - t0) / 1e6:.3f} ms")
for st in sorted({r["Stream_Id"] for r in rows}):
    sel = [r for r in rows if r["Stream_Id"] == st and t0 <= r["s"] < t1]
    if not sel:
        continue
    print(f"--- stream {st}: {len(sel)} launches, busy {sum(r['e'] - r['s'] for r in sel) / 1e6:.3f} ms")
    prev = None
    for r in sel:
        gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
        print(f"  +{(r['s'] - t0) / 1e6:8.3f} ms  gap {gap:8.1f} us  dur {(r['e'] - r['s']) / 1e3:9.1f} us  grid {r['Grid_Size_X']:>8}  {short(r['Kernel_Name'])}")
        prev = r["e"]
