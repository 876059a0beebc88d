"""Kernel-level GB/s of the HBM-bound helpers: average kernel durations from a `rocprofv3 --kernel-trace --stats` run of
tools/bench_aux.py (host-side timings of those calls include ~0.1 ms of Python per call, which dominates kernels of 10-100 us)
over the ALGORITHMIC bytes of each call, against the 8 TB/s HBM3E peak (MI355X_MICROARCH.md).
usage: python tools/aux_kernel_gbps.py <kernel_stats.csv> <out.json>"""
import csv, json, sys

PEAK = 8000.0
VALU_PEAK_TFLOPS = 157.3                           # fp32 vector peak, MI355X (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 32 lanes x 2 (packed) x 2 FLOP x 2.4 GHz)
# kernels whose roofline is NOT HBM: (algorithmic fp32 multiply-adds per launch, why)
COMPUTE_BOUND = {}
n, T, H, W = 50, 16, 112, 112                      # config 3's hallucinator batch: 50 clips 112x112x16
nel = 64 * 3 * 147 + 64 + 128 * 64 * 147 + 128 + 128 * 128 * 147 + 128 + 50 * 128 + 50          # one ConvNet3D gradient list
SPEC = {
    "hal_fwd_kernel": ("vd_hallucinator_fwd", n * (3 * H * W + T * H * W + 3 * T * H * W) * 4),
    "hal_bwd_fused_kernel": ("vd_hallucinator_bwd (round 5, one kernel: upstream gradient, dynamic and static read once; g_dyn + g_stat written; "
                             "bound by vector-ALU / LDS issue -- 162 multiply-adds per pixel and frame --, not by HBM)",
                             n * (3 * T * H * W + 2 * T * H * W + 2 * 3 * H * W) * 4),
    "hal_bwd_data_kernel": ("vd_hallucinator_bwd (data half: upstream gradient read once, g_dyn + g_stat written)", n * (3 * T * H * W + T * H * W + 3 * H * W) * 4),
    "hal_bwd_param_kernel": ("vd_hallucinator_bwd (parameter half: upstream gradient + dynamic + static read)", n * (3 * T * H * W + T * H * W + 3 * H * W) * 4),
    "match_rows_fwd_multi_kernel": ("vd_match_rows_fwd_multi (match_loss forward, one gradient list pair)", 2 * nel * 4),
    "match_rows_bwd_multi_kernel": ("vd_match_rows_bwd_multi", 3 * nel * 4),
    "sgd_momentum_kernel": ("vd_sgd_momentum (50 clips 112x112x16; x, buf, g read, x, buf written)", n * T * 3 * H * W * 20),
    "pix2rows_kernel": ("vd_pix2rows (512 clips -> f16 pixel rows)", 512 * (T * 3 * H * W * 4 + T * 3 * H * 120 * 2)),
    "frames_normalize_quad_kernel": ("vd_frames_normalize (256 clips)", 256 * T * 3 * H * W * (1 + 4)),
    "dm_loss_kernel": ("vd_dm_loss (50 classes x (64 + 1) feature rows of 2048, gradient rows written)", 50 * 65 * 2048 * 4 + 50 * 2048 * 4),
}
# the hallucinator backward is bound by vector-ALU / LDS issue, not by HBM (round 5: removing its atomics, barriers, staging or load latency
# changes nothing; removing arithmetic does): per pixel and frame 81 multiply-adds of the data gradient + 81 of the dynamic channel's
# weight gradient on the vector ALU (the static channels' 243 weight gradients run on the fp32 matrix instruction).  Its roofline is
# the fp32 vector peak; the GB/s figure is kept for continuity.
COMPUTE_BOUND["hal_bwd_fused_kernel"] = (n * T * H * W * 162, "162 fp32 multiply-adds per pixel and frame on the vector ALU (scatter-form data "
                                         "gradient 81 + dynamic-channel weight gradient 81); ~350 instructions issued per pixel and frame incl. LDS reads, "
                                         "v_readlane of the 81 scalar weights, address arithmetic")
COMPUTE_BOUND["hal_fwd_kernel"] = (n * T * H * W * (81 + 243 // T + 3), "per output pixel and frame: 81 multiply-adds of the dynamic channel + the static image's "
                                   "three 2-D sums amortised over the frames")
out = {}
for r in csv.DictReader(open(sys.argv[1])):
    key = r["Name"].split("(")[0]
    if key in SPEC:
        label, byt = SPEC[key]
        us = float(r["AverageNs"]) / 1e3
        out[key] = {"entry": label, "calls": int(r["Calls"]), "avg_us": us, "algorithmic_bytes": byt, "GBps": byt / us / 1e3,
                    "frac_of_8TBps": byt / us / 1e3 / PEAK}
        if key in COMPUTE_BOUND:
            fma, why = COMPUTE_BOUND[key]
            tf = 2.0 * fma / (us * 1e-6) / 1e12
            out[key].update({"bound": "valu", "algorithmic_fp32_fma": fma, "TFLOPs_fp32": tf, "frac_of_valu_peak": tf / VALU_PEAK_TFLOPS,
                             "valu_peak_TFLOPs": VALU_PEAK_TFLOPS, "why": why})
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print("%-32s %8.1f us  %7.0f GB/s  %4.1f %% of HBM peak%s" % (k, v["avg_us"], v["GBps"], 100 * v["frac_of_8TBps"],
          ("   | fp32 vector ALU: %.1f TFLOP/s = %.1f %% of %.0f" % (v["TFLOPs_fp32"], 100 * v["frac_of_valu_peak"], VALU_PEAK_TFLOPS)) if "bound" in v else ""))
