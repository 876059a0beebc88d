"""The C-ABI shared library loads and exports every symbol include/vd_hip.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|void|char\*)\s+(vd_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from video_distillation_amd import hip
    assert declared_functions() == sorted(hip.EXPORTS)


def test_library_exports_every_declared_symbol():
    from video_distillation_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.vd_abi_version.restype = ctypes.c_int
    assert lib.vd_abi_version() == 5


def test_argument_errors_are_reported_before_any_device_call():
    """Entry points validate their arguments first (codes -1 / -2, include/vd_hip.h) -- callable without a GPU."""
    from video_distillation_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    assert lib.vd_conv_mfma(None, None) == -1
    assert lib.vd_conv_mfma_multi(None, 2, None) == -1
    a, b = hip.VdConvParams(), hip.VdConvParams()
    for p, mtw in ((a, 4), (b, 2)):        # two programs of different tile shapes do not share an instantiation
        p.prec, p.NT, p.MW, p.MTW, p.S, p.CC, p.ncl, p.lds_plane_bytes = 3, 1, 4, mtw, 8, 1, 1, 1024
    arr = (ctypes.POINTER(hip.VdConvParams) * 2)(ctypes.pointer(a), ctypes.pointer(b))
    assert lib.vd_conv_mfma_multi(arr, 0, None) == -1 and lib.vd_conv_mfma_multi(arr, 5, None) == -1
    assert lib.vd_conv_mfma_multi(arr, 2, None) == -2
    # a plain and an accumulating (atomic / select) program of the SAME tile shape do not share a launch either: the plain one
    # would run under the second-order instantiation (header: "all plain or all accumulating; -2 otherwise")
    c, d = hip.VdConvParams(), hip.VdConvParams()
    for p in (c, d):
        p.prec, p.NT, p.MW, p.MTW, p.S, p.CC, p.ncl, p.lds_plane_bytes = 2, 1, 4, 4, 8, 1, 1, 1024
    d.atomic = 1
    arr2 = (ctypes.POINTER(hip.VdConvParams) * 2)(ctypes.pointer(c), ctypes.pointer(d))
    assert lib.vd_conv_mfma_multi(arr2, 2, None) == -2
    d.atomic, d.select = 0, 1
    assert lib.vd_conv_mfma_multi(arr2, 2, None) == -2


def test_params_struct_layout_matches_header():
    """ctypes mirror of VdConvParams: same field order as the C struct (names must appear in the
    header in the same sequence) and the size hipcc computes for it."""
    from video_distillation_amd import hip
    text = open(os.path.join(ROOT, "include", "vd_hip.h")).read()
    body = text[text.index("typedef struct VdConvParams {") + len("typedef struct VdConvParams {"):text.index("} VdConvParams;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        decl = re.sub(r"^(const\s+)?[A-Za-z_0-9]+\s*\*?\s*", "", decl, count=1)
        names += [re.sub(r"\[.*\]", "", n.strip().lstrip("*")) for n in decl.split(",")]
    assert names == [f[0] for f in hip.VdConvParams._fields_]
    assert ctypes.sizeof(hip.VdConvParams) % 8 == 0


def test_product_has_no_cpu_fallback():
    import torch
    from video_distillation_amd import networks, utils
    net = networks.ConvNet3D(3, 5, 128, 3, 'relu', 'none', 'maxpooling', 8, (64, 64))
    with pytest.raises(RuntimeError):
        net.embed(torch.zeros(1, 8, 3, 64, 64))
    with pytest.raises(RuntimeError):
        utils.Conv3DNet()(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 1, 8, 8))
    # nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "video_distillation_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_library_carries_the_hash_of_its_sources_and_a_stale_one_is_refused(tmp_path, monkeypatch):
    """Round 6: the library is stamped with the sha256 of the kernel sources + header it was built from (csrc/stamp.cpp);
    ``hip.build`` / ``hip.lib`` decide staleness by that stamp, not by mtimes -- a library built from touched sources is
    rebuilt, and refused when it cannot be."""
    import shutil
    import subprocess
    from video_distillation_amd import hip
    hip.build()
    want = hip.sources_hash()
    assert hip.library_stamp() == want and len(want) == 16
    lib = ctypes.CDLL(hip.LIB_PATH)
    lib.vd_sources_hash.restype = ctypes.c_char_p
    assert lib.vd_sources_hash().decode() == hip.STAMP_PREFIX + want
    # a copy of the library next to "touched" sources (one byte appended to a copy of a kernel file): the stamp no longer matches
    stale = tmp_path / "libvd_hip.so"
    shutil.copy(hip.LIB_PATH, stale)
    src = tmp_path / "aux_kernels.hip"
    shutil.copy(hip.SOURCES[1], src)
    with open(src, "a") as f:
        f.write("\n// touched\n")
    monkeypatch.setattr(hip, "SOURCES", [hip.SOURCES[0], str(src)] + hip.SOURCES[2:])
    monkeypatch.setattr(hip, "LIB_PATH", str(stale))
    monkeypatch.setattr(hip, "_lib", None)
    assert hip.sources_hash() != want and hip.library_stamp(str(stale)) == want

    def no_compiler(*a, **k):
        raise subprocess.CalledProcessError(127, "hipcc")
    monkeypatch.setattr(hip, "build", no_compiler)
    with pytest.raises(RuntimeError, match="built from other sources"):
        hip.lib()
