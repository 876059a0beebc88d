"""Host logic of bench.py that needs no GPU: the stamped PMC traffic figure."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_FILE = "r06_pmc_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")) else "r05_pmc_traffic.json"


def test_pmc_traffic_is_quoted_only_for_the_stamped_kernel_sources(monkeypatch):
    """``roofline.traffic`` travels in a file (counters cannot be read inside the timed run); the file carries the sha256 of the
    kernel sources + ABI header it was measured on, and bench.py quotes it only while that equals the checkout's own hash and
    the launch shape is the file's -- otherwise null with the reason (round 4 quoted a file five commits old)."""
    import bench
    from video_distillation_amd import hip
    doc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
    stamp = doc["_source"]["kernel_sources_sha256_16"]
    rec = doc["conv1_fwd_f16"]
    assert len(stamp) == 16 and rec["clips_per_launch"] == 3200
    if stamp == hip.sources_hash():          # the committed file belongs to the committed sources: quoted
        val, src = bench.pmc_traffic("conv1_fwd_f16", 3200)
        assert val == rec["hbm_bytes_per_launch"] and "measured, unscaled" in src and stamp in src
    else:                                    # a kernel was touched since: refused until tools/final_pmc.sh has re-measured it on a GPU box
        val, src = bench.pmc_traffic("conv1_fwd_f16", 3200)
        assert val is None and "other kernel sources" in src
    monkeypatch.setattr(hip, "sources_hash", lambda: stamp)            # (from here on: as if the file were current)
    val, src = bench.pmc_traffic("conv1_fwd_f16", 400)                 # another launch shape (an 8-rank job): not rescaled, refused
    assert val is None and "not quoted" in src
    monkeypatch.setattr(hip, "sources_hash", lambda: "0" * 16)         # any other kernel sources
    val, src = bench.pmc_traffic("conv1_fwd_f16", 3200)
    assert val is None and "other kernel sources" in src and "re-measure" in src


def test_an_n_rank_line_without_proof_of_its_ranks_has_no_value():
    """``bench.py --gpus N`` withholds ``value`` unless the line proves N ranks: ranks_seen (all-reduce of ones), rccl.nranks
    (ncclCommCount / the process group's size) and the per-rank lists must all say N (reference mechanism replaced:
    nn.DataParallel, utils.py:615-623 -- there a missing device shows up as an exception, here it must not show up as a number)."""
    import bench
    good = {"n_gpus": 8, "value": 230.0, "value_median": 231.0, "ranks_seen": 8, "clips_per_step": [400] * 8,
            "rccl": {"version": "2.26.6", "nranks": 8}}
    assert bench.refuse_unproven(dict(good), 8) is None
    assert bench.refuse_unproven({"n_gpus": 1, "value": 31.0, "ranks_seen": 1, "clips_per_step": [3200], "rccl": None}, 1) is None
    for broken, word in (({"ranks_seen": 7}, "ranks_seen"), ({"rccl": {"nranks": 1}}, "rccl.nranks"),
                         ({"clips_per_step": [400] * 7}, "clips_per_step"), ({"n_gpus": 4}, "n_gpus")):
        line = dict(good, **broken)
        why = bench.refuse_unproven(line, 8)
        assert why and word in why and line["value"] is None and line["value_median"] is None and line["refused"] == why
    # a gloo group on one device reports no RCCL rank count: the all-reduce of ones is the proof
    assert bench.refuse_unproven(dict(good, rccl={"note": "process group on gloo"}), 8) is None
