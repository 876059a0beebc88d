"""Host-side sanitizers (CPU only): the library's HOST code -- csrc/planner.cpp and the handle / workspace-layout half of
csrc/program.hip -- built with ``g++ -fsanitize=address,undefined`` against a host stand-in for the HIP runtime and stand-ins
for the kernel entry points that touch the full extent of their arguments (tests/native/).  The driver plans every program of
three geometries, then runs vd_embed_* and vd_train_* over workspaces malloc'd at EXACTLY the queried sizes and handed over at
misaligned offsets (0, 1, 8, 200 bytes): a carve-out that is too small or a pointer the layout forgot to round is an
AddressSanitizer / alignment report.  (Round 2's vd_train_step overrun -- the slack of rounding an arbitrary caller pointer
up to 256 bytes was not part of the queried size -- had been found by reading.)  No GPU sanitizer exists on this pool."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def _fnv(data: bytes) -> int:
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def driver():
    out = subprocess.run(["make", "-s", "-C", NATIVE, "asan"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return os.path.join(NATIVE, "asan_driver")


def test_planner_and_handles_are_clean_under_asan_and_ubsan(driver):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([driver, "quick"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-4000:]
    assert "asan driver: ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    lines = out.stdout.splitlines()
    assert sum(l.startswith("embed ") for l in lines) == 8 and sum(l.startswith("train ") for l in lines) == 8
    # the sanitizer build planned the SAME programs as the shipped library (hash per serialised blob)
    from video_distillation_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    checked = 0
    for l in lines:
        f = l.split()
        if f[0] == "fwd":
            T, H, W, layer, prec, hint, n, digest = int(f[1]), int(f[2]), int(f[3]), int(f[4][1:]), int(f[5][4:]), int(f[6][4:]), int(f[7]), f[8]
            blob, nb = ctypes.c_void_p(), ctypes.c_int64()
            assert lib.vd_program_build(layer, T, H, W, prec, hint, ctypes.byref(blob), ctypes.byref(nb)) == 0
        elif f[0] == "dgrad":
            T, H, W, layer, cls, n, digest = int(f[1]), int(f[2]), int(f[3]), int(f[4][1:]), int(f[5][1:]), int(f[6]), f[7]
            blob, nb = ctypes.c_void_p(), ctypes.c_int64()
            assert lib.vd_program_build_dgrad(layer, cls, T, H, W, 64, ctypes.byref(blob), ctypes.byref(nb)) == 0
        elif f[0] == "wgrad":
            T, H, W, layer, nclips, det, n, digest = int(f[1]), int(f[2]), int(f[3]), int(f[4][1:]), int(f[5][1:]), int(f[6][3:]), int(f[7]), f[8]
            blob, nb = ctypes.c_void_p(), ctypes.c_int64()
            block, rep = (ctypes.c_int * 3)(), ctypes.c_int()
            prev = lib.vd_set_deterministic(det)
            try:
                assert lib.vd_program_build_wgrad(layer, T, H, W, nclips, 2, ctypes.byref(blob), ctypes.byref(nb), block, ctypes.byref(rep)) == 0
            finally:
                lib.vd_set_deterministic(prev)
            assert [block[0], block[1], block[2], rep.value] == [int(v) for v in (f[10], f[11], f[12], f[14])]
        else:
            continue
        data = ctypes.string_at(blob, nb.value)
        lib.vd_blob_free.restype = None
        lib.vd_blob_free(blob)
        assert nb.value == n and "%016x" % _fnv(data) == digest, l
        checked += 1
    assert checked >= 3 * (3 * 4 + 9 + 4) - 9          # 3 geometries x (12 forward + 9 input-gradient + 4 weight-gradient blobs)


def test_the_harness_sees_an_overrun(driver):
    """Self-test of the instrument: a training workspace 512 bytes shorter than vd_train_step is told must be reported."""
    out = subprocess.run([driver, "selftest-overrun"], capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in out.stderr and "NOT detected" not in out.stdout
