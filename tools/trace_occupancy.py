"""Device occupancy from a rocprofv3 --kernel-trace csv: busy fraction (union of kernel intervals), mean number of kernels in flight,
and the share of busy time in which fewer than `slots` workgroups were resident (sum over the kernels in flight of min(grid, slots)).
usage: python tools/trace_occupancy.py <kernel_trace.csv> [t0_frac t1_frac | --longest]   (--longest: the longest stretch of
the trace without an idle gap of 3 ms, e.g. the timed steps of a bench run)"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if (len(sys.argv) > 3 and not sys.argv[2].startswith("--")) else (0.0, 1.0)
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
wg = np.array([max(1, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) for r in rows], dtype=np.int64)
names = [r["Kernel_Name"] for r in rows]
T0, T1 = st.min(), en.max()
lo, hi = T0 + (T1 - T0) * f0, T0 + (T1 - T0) * f1
if "--segments" in sys.argv:       # list the busy stretches (no idle gap of 3 ms) of at least 50 ms
    order = np.argsort(st)
    seg_lo, seg_hi, k = st[order[0]], en[order[0]], 0
    segs = []
    for i in order[1:]:
        if st[i] > seg_hi + 3_000_000:
            segs.append((seg_lo, seg_hi, k)); seg_lo, seg_hi, k = st[i], en[i], 0
        seg_hi = max(seg_hi, en[i]); k += 1
    segs.append((seg_lo, seg_hi, k))
    for a, b, k in segs:
        if b - a > 50_000_000:
            print("segment at %.1f ms: %.1f ms, %d kernels  -> fractions %.4f %.4f" % ((a - T0) / 1e6, (b - a) / 1e6, k, (a - T0) / (T1 - T0), (b - T0) / (T1 - T0)))
    sys.exit(0)
if "--longest" in sys.argv:
    order = np.argsort(st)
    best, seg_lo, seg_hi = (0, T0, T1), st[order[0]], en[order[0]]
    for i in order[1:]:
        if st[i] > seg_hi + 3_000_000:
            if seg_hi - seg_lo > best[0]:
                best = (seg_hi - seg_lo, seg_lo, seg_hi)
            seg_lo, seg_hi = st[i], en[i]
        seg_hi = max(seg_hi, en[i])
    if seg_hi - seg_lo > best[0]:
        best = (seg_hi - seg_lo, seg_lo, seg_hi)
    lo, hi = best[1], best[2]
sel = (en > lo) & (st < hi)
st, en, wg = np.clip(st[sel], lo, hi), np.clip(en[sel], lo, hi), wg[sel]
names = [n for n, s in zip(names, sel) if s]
ev = sorted([(t, +1, w) for t, w in zip(st, wg)] + [(t, -1, w) for t, w in zip(en, wg)])
busy = inflight_t = under = 0.0
cur_k = cur_w = 0
hist = {}
prev = ev[0][0]
SLOTS = 512
for t, d, w in ev:
    dt = t - prev
    if cur_k > 0 and dt > 0:
        busy += dt; inflight_t += dt * cur_k
        if cur_w < SLOTS:
            under += dt
        b = min(cur_w, 4096) // 128 * 128
        hist[b] = hist.get(b, 0) + dt
    cur_k += d; cur_w += d * w
    prev = t
span = hi - lo
print("window %.1f ms, busy %.1f ms (%.1f %%), kernels in flight while busy %.2f, busy time with < %d workgroups launched-and-unfinished: %.1f %%" % (
    span / 1e6, busy / 1e6, 100 * busy / span, inflight_t / max(busy, 1), SLOTS, 100 * under / max(busy, 1)))
print("busy time by total workgroups of the kernels in flight (bucket of 128):")
for b in sorted(hist):
    if hist[b] / busy > 0.01:
        print("  >= %5d: %5.1f %%" % (b, 100 * hist[b] / busy))
dur = {}
for n, a, b in zip(names, st, en):
    dur[n] = dur.get(n, 0) + (b - a)
print("kernel time by name (sum of durations / window):")
for n, d in sorted(dur.items(), key=lambda kv: -kv[1])[:14]:
    print("  %6.1f %%  %s" % (100 * d / span, n[:110]))
