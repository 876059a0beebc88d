"""ctypes binding of libvd_hip.so (C ABI declared in include/vd_hip.h).

PyTorch is used only as the owner of device memory and streams: every call passes
``tensor.data_ptr()`` and the current HIP stream handle.  There is deliberately NO fallback:
if the shared library is missing or a kernel launch fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvd_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("conv_mfma.hip", "aux_kernels.hip", "program.hip", "planner.cpp", "comm.cpp")]
STAMP_SOURCE = os.path.join(_HERE, "csrc", "stamp.cpp")      # vd_sources_hash(): compiled on every link with the hash of SOURCES + header
HEADER = os.path.join(_HERE, "..", "include", "vd_hip.h")

PREC = {"bf16": 0, "f16": 1, "bf16x3": 2, "f16x3": 3, "f16c8": 4}      # f16c8: fp16 + fp8 corrections (the real side's last level only)
EXPORTS = ("vd_abi_version", "vd_conv_mfma", "vd_conv_mfma_multi", "vd_conv0_breg", "vd_pack_weights", "vd_round_operand", "vd_pix2rows", "vd_unpool_relu_bwd", "vd_absmax_scale", "vd_dm_loss",
           "vd_group_sum", "vd_sgd_momentum", "vd_hallucinator_fwd", "vd_hallucinator_bwd", "vd_match_rows_fwd", "vd_match_rows_bwd", "vd_match_rows_fwd_multi", "vd_match_rows_bwd_multi", "vd_head_fwd", "vd_clip_minor_cl", "vd_clip_minor_pix", "vd_pack_dy", "vd_bias_grad", "vd_bias_grad_pooled", "vd_standardize", "vd_head_train_fwd", "vd_ce_loss", "vd_head_train_bwd", "vd_head_second_order", "vd_resplit_slots", "vd_program_load", "vd_program_pack_weights",
           "vd_program_run", "vd_program_info", "vd_program_free",
           "vd_sgd_momentum_wd", "vd_frames_normalize", "vd_replica_sum", "vd_pack_weights_dither", "vd_unpool_relu_bwd_packed", "vd_mfma_peak", "vd_program_build", "vd_program_build_dgrad", "vd_program_build_wgrad", "vd_program_run_wgrad", "vd_train_create", "vd_train_workspace_bytes", "vd_train_step", "vd_train_free", "vd_blob_free", "vd_embed_create", "vd_embed_create_ex", "vd_embed_argmax_bytes", "vd_embed_backward_workspace_bytes",
           "vd_embed_forward_keep", "vd_embed_backward", "vd_program_run_scaled", "vd_embed_num_features",
           "vd_embed_workspace_bytes", "vd_embed_set_weights", "vd_embed_forward", "vd_embed_free",
           "vd_comm_unique_id", "vd_comm_create", "vd_comm_size", "vd_comm_version", "vd_comm_rank", "vd_comm_allreduce_f32", "vd_comm_allgather_f32",
           "vd_comm_free",
           "vd_bias_grad_pooled_scratch_floats", "vd_bias_grad_pooled_ordered", "vd_standardize_ordered", "vd_head_train_bwd_ordered",
           "vd_set_deterministic", "vd_get_deterministic", "vd_pack_weights_c8", "vd_pack_weights_multi",
           "vd_split_scaled", "vd_scale_combine", "vd_sources_hash")
F16X3_WSHIFT = 8          # include/vd_hip.h VD_F16X3_WSHIFT: packed fp16 hi+lo weights are W x 2^8, undone in the programs' epilogues


class VdConvParams(ctypes.Structure):
    _fields_ = [
        ("src", ctypes.c_void_p), ("src_plane_stride4", ctypes.c_int64),
        ("src_clip_stride4", ctypes.c_int64), ("src_chunk_stride4", ctypes.c_int64),
        ("wpk", ctypes.c_void_p), ("w_plane_stride", ctypes.c_int64), ("w_box_stride", ctypes.c_int64),
        ("bias", ctypes.c_void_p),
        ("dst", ctypes.c_void_p), ("dst_plane_stride", ctypes.c_int64),
        ("argmax", ctypes.c_void_p), ("col_off", ctypes.c_void_p), ("out_scale", ctypes.c_void_p),
        ("type_desc", ctypes.c_void_p), ("tables", ctypes.c_void_p), ("boxes", ctypes.c_void_p),
        ("gather", ctypes.c_void_p), ("gather_stride", ctypes.c_int64), ("zero_slot", ctypes.c_void_p),
        ("nbox", ctypes.c_int32), ("nclips", ctypes.c_int32), ("ncl", ctypes.c_int32),
        ("CC", ctypes.c_int32), ("S", ctypes.c_int32), ("NT", ctypes.c_int32), ("MW", ctypes.c_int32), ("MTW", ctypes.c_int32),
        ("epi", ctypes.c_int32), ("pool_t", ctypes.c_int32), ("relu", ctypes.c_int32),
        ("n_out", ctypes.c_int32), ("n_stride", ctypes.c_int32),
        ("out_clip_stride", ctypes.c_int64),
        ("out_chunk_stride", ctypes.c_int32), ("out_t_stride", ctypes.c_int32),
        ("lds_plane_bytes", ctypes.c_int32), ("prec", ctypes.c_int32), ("dbg", ctypes.c_int32), ("ntypes", ctypes.c_int32), ("tab_ofs", ctypes.c_int32 * 3), ("atomic", ctypes.c_int32), ("select", ctypes.c_int32), ("src_split_cc", ctypes.c_int32), ("src_split_off4", ctypes.c_int64), ("NTW", ctypes.c_int32), ("clip_index", ctypes.c_void_p), ("mt_valid", ctypes.c_int32), ("persist", ctypes.c_int32), ("stamps", ctypes.c_void_p),
        ("w_set_clips", ctypes.c_int32), ("replica_stride", ctypes.c_int32), ("emit_lo", ctypes.c_int32), ("src_planes", ctypes.c_int32), ("src_rows", ctypes.c_int32), ("pair_flip", ctypes.c_int32), ("range_stats", ctypes.c_void_p),
    ]


class VdMatchSeg(ctypes.Structure):
    _fields_ = [("gr", ctypes.c_void_p), ("gs", ctypes.c_void_p), ("g", ctypes.c_void_p), ("rows", ctypes.c_int64),
                ("len", ctypes.c_int32), ("reserved", ctypes.c_int32)]


VD_PACK_MAX = 24


class VdPackSeg(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("widx", ctypes.c_void_p), ("n", ctypes.c_int64), ("out_hi", ctypes.c_void_p),
                ("out_lo", ctypes.c_void_p), ("prec", ctypes.c_int32), ("first_block", ctypes.c_int32)]


class VdPackBatch(ctypes.Structure):
    _fields_ = [("nseg", ctypes.c_int32), ("reserved", ctypes.c_int32), ("seg", VdPackSeg * VD_PACK_MAX)]


class VdMatchBatch(ctypes.Structure):
    _fields_ = [("nseg", ctypes.c_int32), ("reserved", ctypes.c_int32), ("seg", VdMatchSeg * 16)]


def build(force: bool = False, verbose: bool = False, debug_hooks: bool = False) -> str:
    """Compile the HIP sources for gfx950 into libvd_hip.so (in-tree).  ``debug_hooks`` builds the
    variant libvd_hip_dbg.so with the ablation / timing hooks of VdConvParams.dbg compiled in
    (-DVD_DBG_HOOKS=1; used by tools/ablate.py and tools/stamps.py via VD_LIB_VARIANT=dbg).

    Staleness is decided by CONTENT, not by mtime (round 6): the library carries the sha256 of the sources it was built from
    (csrc/stamp.cpp, ``vd_sources_hash``) and is rebuilt when that differs from ``sources_hash()`` of this checkout; every
    object file has a side file with the hash of its source + the header."""
    out = LIB_PATH.replace(".so", "_dbg.so") if debug_hooks else LIB_PATH
    if not force and library_stamp(out) == sources_hash():
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # one object per source (rebuilt only when that source or the header changed), compiled concurrently, then one link
    objdir = os.path.join(_HERE, "csrc", "build_dbg" if debug_hooks else "build")
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time (several ranks / test workers may find the library stale together): an exclusive lock on the object
    # directory; whoever waited finds the library fresh and returns
    import fcntl
    lock = open(os.path.join(objdir, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        return _build_locked(out, objdir, hipcc, force, verbose, debug_hooks)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _file_hash(*paths: str) -> str:
    import hashlib
    h = hashlib.sha256()
    for path in paths:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _build_locked(out: str, objdir: str, hipcc: str, force: bool, verbose: bool, debug_hooks: bool) -> str:
    want = sources_hash()
    if not force and library_stamp(out) == want:
        return out
    flags = ["-O3", "--offload-arch=gfx950", "-fPIC"] + (["-DVD_DBG_HOOKS=1"] if debug_hooks else [])
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        side = obj + ".srchash"
        objs.append(obj)
        key = _file_hash(src, HEADER)
        have = open(side).read().strip() if os.path.exists(side) and os.path.exists(obj) else None
        if force or have != key:
            if os.path.exists(side):
                os.remove(side)
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd), side, key))
    for cmd, pr, side, key in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        with open(side, "w") as f:
            f.write(key)
    stamp_obj = os.path.join(objdir, "stamp.cpp.o")
    cmd = [hipcc, "-O2", "-fPIC", "-DVD_SOURCES_HASH=\"%s%s\"" % (STAMP_PREFIX, want), "-c", STAMP_SOURCE, "-o", stamp_obj]
    subprocess.run(cmd, check=True)
    tmp = out + ".tmp.%d" % os.getpid()       # (linked beside the target and renamed: a reader never maps a half-written library)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + [stamp_obj, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)
    return out


STAMP_PREFIX = "VD_SOURCES_HASH="


def library_stamp(path: str = None) -> Optional[str]:
    """The sources hash a built library carries (read from the file's bytes, no dlopen: a stale library must not be mapped into
    the process that is about to rebuild it), or None if the file is missing or unstamped."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    import mmap
    with open(path, "rb") as f:
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
            i = m.find(STAMP_PREFIX.encode())
            if i < 0:
                return None
            return m[i + len(STAMP_PREFIX):i + len(STAMP_PREFIX) + 16].decode("ascii", "replace")


def sources_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and the ABI header: what measurement files that are carried from one
    run to the next (profiles/rNN_pmc_traffic.json -> bench.py's ``roofline.traffic``) are stamped with, so that a figure
    measured on other kernels is refused instead of quoted."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(SOURCES) + [HEADER]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("VD_LIB_PATH", LIB_PATH)           # (measurement tools: A/B of two builds on one box)
        if os.environ.get("VD_LIB_VARIANT") == "dbg":      # profiling tools: the build with the dbg hooks compiled in
            path = build(debug_hooks=True)
        if not os.path.exists(path):
            raise RuntimeError(
                "libvd_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU or eager fallback for the HIP path)" % path)
        if "VD_LIB_PATH" not in os.environ and os.environ.get("VD_LIB_VARIANT") != "dbg":
            # the numbers of a run come from THIS file: it must be the build of this checkout's kernel sources (the stamp
            # compiled into it, not its mtime, says so).  A stale library is rebuilt when a compiler is there, refused otherwise.
            want = sources_hash()
            if library_stamp(path) != want:
                try:
                    build()
                except (OSError, subprocess.CalledProcessError) as e:
                    raise RuntimeError("libvd_hip.so at %s was built from other sources (stamp %s, checkout %s) and could not be "
                                       "rebuilt: %s" % (path, library_stamp(path), want, e))
                if library_stamp(path) != want:
                    raise RuntimeError("libvd_hip.so at %s carries stamp %s, the checkout's kernel sources hash to %s"
                                       % (path, library_stamp(path), want))
        L = ctypes.CDLL(path)
        for name in EXPORTS:
            if not hasattr(L, name):
                if "VD_LIB_PATH" in os.environ:      # an older build loaded on purpose for an A/B measurement
                    continue
                raise RuntimeError("libvd_hip.so does not export %s" % name)
            getattr(L, name).restype = {"vd_program_info": ctypes.c_int64, "vd_program_free": None, "vd_blob_free": None, "vd_embed_free": None, "vd_train_free": None, "vd_comm_free": None, "vd_train_workspace_bytes": ctypes.c_int64, "vd_bias_grad_pooled_scratch_floats": ctypes.c_int64,
                                        "vd_embed_num_features": ctypes.c_int64, "vd_embed_workspace_bytes": ctypes.c_int64,
                                        "vd_embed_argmax_bytes": ctypes.c_int64, "vd_embed_backward_workspace_bytes": ctypes.c_int64}.get(name, ctypes.c_int)
        if L.vd_abi_version() != 5:
            raise RuntimeError("libvd_hip.so ABI version mismatch")
        if hasattr(L, "vd_sources_hash"):
            L.vd_sources_hash.restype = ctypes.c_char_p
        _lib = L
    return _lib


def loaded_stamp() -> str:
    """Sources hash compiled into the library this process runs on (bench.py and smoke() print it)."""
    v = lib().vd_sources_hash()
    v = v.decode() if isinstance(v, bytes) else str(v)
    return v[len(STAMP_PREFIX):] if v.startswith(STAMP_PREFIX) else v


def deterministic() -> bool:
    """Whether accumulations run in a fixed order (bitwise reproducible training step): the library's process-wide
    switch, initialised from VD_DETERMINISTIC."""
    return bool(lib().vd_get_deterministic())


def set_deterministic(on: bool) -> bool:
    """Switch the fixed-order accumulation mode; engines / programs created afterwards follow it.  Returns the previous value."""
    return bool(lib().vd_set_deterministic(int(bool(on))))


def stream_ptr(device=None) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError("%s failed with code %d" % (what, code))


def ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def is_x3(prec: int) -> bool:
    return prec >= 2


class Comm:
    """An RCCL communicator behind the C ABI (vd_comm_*, include/vd_hip.h): created over the ranks of the default torch.distributed
    group (rank 0's 128-byte id travels through ``broadcast_object_list``), collectives issued on the caller's CURRENT HIP stream.
    Used where the exchange goes through the library's own entry points instead of torch's process group (bench.py's
    ``--exchange allreduce`` leg and the rccl block of its JSON line)."""

    def __init__(self, rank: int, world: int):
        import torch.distributed as dist
        L = lib()
        ident = (ctypes.c_char * 128)()
        if rank == 0:
            check(L.vd_comm_unique_id(ident), "vd_comm_unique_id")
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (ctypes.c_char * 128).from_buffer_copy(box[0])
        self._c = ctypes.c_void_p()
        check(L.vd_comm_create(ident, world, rank, ctypes.byref(self._c)), "vd_comm_create")
        self.rank, self.world = rank, world

    def size(self) -> int:
        return int(lib().vd_comm_size(self._c))

    @staticmethod
    def version() -> Optional[int]:
        v = ctypes.c_int(0)
        return int(v.value) if lib().vd_comm_version(ctypes.byref(v)) == 0 else None

    def all_reduce(self, t: torch.Tensor) -> None:
        assert t.dtype == torch.float32 and t.is_contiguous()
        check(lib().vd_comm_allreduce_f32(self._c, ptr(t), ptr(t), ctypes.c_int64(t.numel()), stream_ptr(t.device)), "vd_comm_allreduce_f32")

    def free(self) -> None:
        if self._c:
            lib().vd_comm_free(self._c)
            self._c = ctypes.c_void_p()
