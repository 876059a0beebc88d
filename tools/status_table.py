"""The numbers of DESIGN.md section 8 from the committed bench lines under profiles/ (python tools/status_table.py [rNN])."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def load(name):
    path = os.path.join(P, "%s_%s.json" % (tag, name))
    return json.load(open(path)) if os.path.exists(path) and os.path.getsize(path) > 2 else None


def line(name, label):
    d = load(name)
    if not d:
        return
    r = d.get("roofline") or {}
    extra = ""
    if "step_frac_of_mfma_peak" in d:
        extra += "; step %.3f of the spec peak" % d["step_frac_of_mfma_peak"]
    if d.get("sustained"):
        extra += "; sustained %.2f" % d["sustained"]["value"]
    print("| %s | **%.2f %s** (%.2f ms; median %.2f ms)%s; dominant program `%s`: %.0f TFLOP/s = %.3f of spec%s; library %s |" % (
        label, d["value"], "it/s" if "mtt" in name else "steps/s", d["ms_per_step"], d["ms_per_step_median"], extra,
        (r.get("kernel") or "").split("'")[1] if "'" in (r.get("kernel") or "") else "-", r.get("achieved", 0), r.get("frac", 0),
        (", %.2f of the measured %.0f" % (r["achieved"] / r["peak_measured"], r["peak_measured"])) if r.get("peak_measured") else "",
        (d.get("library") or {}).get("sources_hash")))


d = load("bench_1gpu")
if d:
    r = d["roofline"]
    print("| DM, C = 50, 64 real + 1 syn clips 112x112x16 per class, fresh net per step, shipped mode | **%.2f steps/s** (%.2f ms; median %.2f; sustained %s) |" % (
        d["value"], d["ms_per_step"], d["ms_per_step_median"], ("%.2f" % d["sustained"]["value"]) if d.get("sustained") else "-"))
    print("| the same run: `fast_mode` / `parity_mode` | %s / %s steps/s |" % tuple(
        ("%.2f" % d[k]["value"]) if d.get(k) else "-" for k in ("fast_mode", "parity_mode")))
    print("| algorithmic FLOP / step time | 36.31 TFLOP -> %.0f TFLOP/s = %.3f of the 2.5 PFLOP/s spec peak |" % (36.31 / d["ms_per_step"] * 1e3, 36.31 / d["ms_per_step"] * 1e3 / 2500))
    print("| dominant kernel (%s) | in the step: %.2f ms mean launch -> %.0f TFLOP/s = **%.3f of spec**, %.2f of the measured %.0f; alone: %s |" % (
        r["kernel"], r["mean_launch_ms"], r["achieved"], r["frac"], r["achieved"] / r["peak_measured"], r["peak_measured"],
        json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.get("alone") or {}).items() if k != "note"})))
    print("| `roofline.traffic` | %s (%s) |" % (r.get("traffic"), (r.get("traffic_source") or "")[:160]))
    cb = d.get("cpu_baseline") or {}
    print("| CPU baseline | %.4f steps/s on %s threads (%s); GPU-vs-CPU loss %.1e |" % (cb.get("value", 0), cb.get("cores"), (cb.get("sample") or "")[:80], cb.get("loss_rel_err_vs_gpu", 0)))
    print("| eval (a smoke) | top-1 per seed %s |" % (d.get("eval") or {}).get("top1_per_seed"))
line("bench_s2d", "s2d (config 3)")
line("bench_config1_shape", "config 1's shape on the GPU (64x64x8)")
line("bench_dc", "gradient matching (config 4), scaled fp16 pairs")
line("bench_dc_bf16x3", "... same box, bf16 pairs (VD_PREC_MATCH=bf16x3)")
line("bench_mtt", "MTT+Ours (config 5), scaled fp16 pairs")
line("bench_mtt_bf16x3", "... same box, bf16 pairs (VD_PREC_MATCH=bf16x3)")
for f in ("train_step.txt", "train_step_deterministic.txt", "hal_bwd.txt", "rank_proxy.txt", "l0_kernel_ab.txt"):
    path = os.path.join(P, "%s_%s" % (tag, f))
    if os.path.exists(path):
        print("| `%s` | %s |" % (os.path.basename(path), " / ".join(l.strip()[:150] for l in open(path).read().strip().splitlines()[-3:])))
