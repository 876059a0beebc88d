"""The C++ planner (csrc/planner.cpp, vd_program_build) against the Python planner: byte-identical serialised forward
programs for the benchmark geometries, an odd one, both operand-precision families and small-batch hints.  No GPU needed:
planning is host code."""
import ctypes
import os

import pytest

from video_distillation_amd import hip, plan

LIB = hip.LIB_PATH


def _cpp_blob(lib, layer, geo, prec, hint):
    blob, n = ctypes.c_void_p(), ctypes.c_int64()
    rc = lib.vd_program_build(layer, geo.frames, geo.height, geo.width, hip.PREC[prec], hint or 0, ctypes.byref(blob), ctypes.byref(n))
    assert rc == 0, rc
    try:
        return ctypes.string_at(blob, n.value)
    finally:
        lib.vd_blob_free(blob)


@pytest.mark.skipif(not os.path.exists(LIB), reason="libvd_hip.so not built")
@pytest.mark.parametrize("dims,prec,hint", [((16, 112, 112), "f16", None), ((16, 112, 112), "f16x3", 64), ((8, 64, 64), "f16", None),
                                            ((8, 64, 64), "bf16x3", 8), ((8, 64, 64), "f16x3", 256), ((8, 80, 96), "f16", 4),
                                            ((4, 64, 64), "f16x3", None)])
def test_cpp_planner_emits_the_python_planner_s_programs(dims, prec, hint):
    lib = ctypes.CDLL(LIB)
    lib.vd_blob_free.restype = None
    geo = plan.NetGeometry(*dims)
    x3 = prec.endswith("x3")
    net = plan.plan_network(geo, ntw=2, ntw0=1, balanced=not x3, batch_hint=hint)
    for layer in range(3):
        want = plan.export_program(net["fwd"][layer])
        got = _cpp_blob(lib, layer, geo, prec, hint)
        assert len(got) == len(want), (layer, len(got), len(want))
        assert got == want, "layer %d differs" % layer


@pytest.mark.skipif(not os.path.exists(LIB), reason="libvd_hip.so not built")
def test_cpp_planner_argument_errors():
    lib = ctypes.CDLL(LIB)
    blob, n = ctypes.c_void_p(), ctypes.c_int64()
    assert lib.vd_program_build(3, 16, 112, 112, 1, 0, ctypes.byref(blob), ctypes.byref(n)) == -1
    assert lib.vd_program_build(0, 16, 112, 112, 7, 0, ctypes.byref(blob), ctypes.byref(n)) == -1
    assert lib.vd_program_build(0, 16, 8, 8, 1, 0, ctypes.byref(blob), ctypes.byref(n)) == -2
    assert lib.vd_program_build(0, 16, 112, 112, 1, 0, None, ctypes.byref(n)) == -1


@pytest.mark.skipif(not os.path.exists(LIB), reason="libvd_hip.so not built")
@pytest.mark.parametrize("dims,hint", [((16, 112, 112), None), ((16, 112, 112), 50), ((8, 64, 64), None), ((8, 64, 64), 8), ((8, 80, 96), 4)])
def test_cpp_planner_emits_the_python_planner_s_input_gradient_programs(dims, hint, monkeypatch):
    if hint == 50:       # one combination with the 2 x 2 pixel blocks of rounds 1-3 (both planners read the same switch)
        monkeypatch.setenv("VD_BWD0_WIDE", "0")
    lib = ctypes.CDLL(LIB)
    lib.vd_blob_free.restype = None
    geo = plan.NetGeometry(*dims)
    net = plan.plan_network(geo, ntw=2, ntw0=1, balanced=True, batch_hint=hint)
    assert net["bwd"][0][0].meta["block_w"] == (2 if hint == 50 else 4) and net["bwd"][0][0].n_out == (12 if hint == 50 else 24)
    for layer in range(3):
        for cls, pl in enumerate(net["bwd"][layer]):
            want = plan.export_program(pl)
            blob, n = ctypes.c_void_p(), ctypes.c_int64()
            rc = lib.vd_program_build_dgrad(layer, cls, geo.frames, geo.height, geo.width, hint or 0, ctypes.byref(blob), ctypes.byref(n))
            assert rc == 0, (layer, cls, rc)
            try:
                got = ctypes.string_at(blob, n.value)
            finally:
                lib.vd_blob_free(blob)
            assert got == want, "dgrad layer %d class %d differs" % (layer, cls)
        blob, n = ctypes.c_void_p(), ctypes.c_int64()
        assert lib.vd_program_build_dgrad(layer, len(net["bwd"][layer]), geo.frames, geo.height, geo.width, 0, ctypes.byref(blob), ctypes.byref(n)) == -3


@pytest.mark.skipif(not os.path.exists(LIB), reason="libvd_hip.so not built")
@pytest.mark.parametrize("dims,nclips,planes", [((16, 112, 112), 64, 2), ((16, 112, 112), 64, 1), ((16, 112, 112), 5, 2), ((8, 64, 64), 256, 2),
                                                ((8, 64, 64), 50, 1), ((8, 80, 96), 7, 2), ((4, 64, 64), 3, 1)])
def test_cpp_planner_emits_the_python_planner_s_weight_gradient_programs(dims, nclips, planes):
    """vd_program_build_wgrad == plan.plan_wgrad: the same block of positions, the same accumulation copies, byte-identical
    serialised programs, for every layer of the benchmark geometries and an odd one, both operand-plane counts, batch sizes
    from a ragged handful to 256."""
    lib = ctypes.CDLL(LIB)
    lib.vd_blob_free.restype = None
    geo = plan.NetGeometry(*dims)
    for layer, (cin, cout, t, h, w) in enumerate([d[:5] for d in geo.layer_dims()]):
        pl = plan.plan_wgrad("wgrad%dx%d" % (cin, cout), cin, cout, t, h, w, nclips, planes=planes)
        want = plan.export_program(pl)
        blob, n = ctypes.c_void_p(), ctypes.c_int64()
        block, reps = (ctypes.c_int * 3)(), ctypes.c_int()
        rc = lib.vd_program_build_wgrad(layer, geo.frames, geo.height, geo.width, nclips, planes, ctypes.byref(blob), ctypes.byref(n),
                                        block, ctypes.byref(reps))
        assert rc == 0, rc
        try:
            got = ctypes.string_at(blob, n.value)
        finally:
            lib.vd_blob_free(blob)
        assert tuple(block) == tuple(pl.meta["box"]) and reps.value == pl.meta["replicas"], (layer, tuple(block), pl.meta["box"])
        assert len(got) == len(want) and got == want, "layer %d differs" % layer
    blob, n, block, reps = ctypes.c_void_p(), ctypes.c_int64(), (ctypes.c_int * 3)(), ctypes.c_int()
    assert lib.vd_program_build_wgrad(3, 16, 112, 112, 8, 1, ctypes.byref(blob), ctypes.byref(n), block, ctypes.byref(reps)) == -1
    assert lib.vd_program_build_wgrad(0, 16, 112, 112, 0, 1, ctypes.byref(blob), ctypes.byref(n), block, ctypes.byref(reps)) == -2
    assert lib.vd_program_build_wgrad(0, 16, 112, 112, 8, 3, ctypes.byref(blob), ctypes.byref(n), block, ctypes.byref(reps)) == -2
