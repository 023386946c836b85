"""ctypes binding of ``libpadel_hip.so`` (C-ABI: ``include/padel_hip.h``).

This is the only place the package talks to the GPU.  There is deliberately NO CPU fallback: if
the shared library or a GPU is missing every entry point raises ``EngineUnavailable`` (the product
path must fail loudly rather than silently route through a CPU restatement).
"""
from __future__ import annotations

import ctypes as C
import sys as _sys
import os
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import graph as G

# PADEL_LIB: tuning tools only (A/B runs against a second build of the library, e.g. tools/ab/*.so)
_LIB_PATH = Path(os.environ["PADEL_LIB"]).resolve() if os.environ.get("PADEL_LIB") else Path(__file__).resolve().parent / "libpadel_hip.so"
_lib = None


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    pass


class RangeOverflow(EngineError):
    """An activation of an h2 model (fp16 pairs, |x| <= 65504) did not fit: the results of that call are invalid and
    the caller repeats it on the full-range fp32 (bf16x3) model (yolo.py / trackers/ball_tracker.py do)."""


def fp32_mode() -> str:
    """Arithmetic of the fp32-equivalent path: "h2" (default: fp16 pairs, three MFMA products, csrc/h2_common.h) or
    "bx3" (exact bf16 triples, six products: full fp32 range, the fallback of the former).  PADEL_FP32_MODE overrides."""
    m = os.environ.get("PADEL_FP32_MODE", "h2").lower()
    if m not in ("h2", "bx3"):
        raise ValueError(f"PADEL_FP32_MODE={m!r}: expected 'h2' or 'bx3'")
    return m


class pa_buf_desc(C.Structure):
    _fields_ = [("level", C.c_int32), ("channels", C.c_int32)]


class pa_op_desc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_buf", C.c_int32), ("in_choff", C.c_int32), ("cin", C.c_int32),
                ("out_buf", C.c_int32), ("out_choff", C.c_int32), ("cout", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("res_buf", C.c_int32), ("res_choff", C.c_int32), ("npad", C.c_int32), ("reserved", C.c_int32),
                ("w_off", C.c_int64), ("b_off", C.c_int64), ("flags", C.c_int32), ("pad_", C.c_int32)]


class pa_model_desc(C.Structure):
    _fields_ = [("task", C.c_int32), ("nc", C.c_int32), ("nk", C.c_int32), ("kpt_dim", C.c_int32),
                ("n_bufs", C.c_int32), ("bufs", C.POINTER(pa_buf_desc)),
                ("n_ops", C.c_int32), ("ops", C.POINTER(pa_op_desc)),
                ("head_buf", C.c_int32 * 3), ("in_channels", C.c_int32), ("dtype", C.c_int32)]


class pa_yolo_params(C.Structure):
    _fields_ = [("imgsz", C.c_int32), ("pre_mode", C.c_int32), ("channel_reverse", C.c_int32),
                ("letterbox_auto", C.c_int32), ("conf", C.c_float), ("iou", C.c_float),
                ("max_det", C.c_int32), ("n_classes", C.c_int32), ("classes", C.POINTER(C.c_int32)),
                ("frames_on_device", C.c_int32)]


PRE_LETTERBOX, PRE_PIL_STRETCH = 0, 1

# every symbol include/padel_hip.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "pa_abi_version", "pa_device_count", "pa_engine_create", "pa_engine_destroy", "pa_last_error",
    "pa_engine_synchronize", "pa_device_malloc", "pa_device_free", "pa_memcpy_h2d", "pa_memcpy_d2h",
    "pa_model_create", "pa_model_destroy", "pa_model_set_max_batch", "pa_yolo_infer", "pa_yolo_head_shape",
    "pa_yolo_read_head", "pa_tracknet_infer", "pa_engine_set_profiling", "pa_model_last_profile",
    "pa_model_profile_text", "pa_ball_create", "pa_ball_destroy", "pa_ball_set_background", "pa_ball_feed",
    "pa_ball_locate", "pa_ball_background_from_frames",
    "pa_upload", "pa_engine_set_tuning", "pa_engine_set_timeline_path", "pa_model_plan_bytes",
    "pa_yolo_netin_shape", "pa_yolo_read_netin",
    "pa_comm_unique_id", "pa_engine_comm_init", "pa_engine_comm_destroy", "pa_engine_bcast_weights",
    "pa_engine_bcast", "pa_engine_allreduce_max", "pa_engine_gather_sizes", "pa_engine_gather",
    "pa_bytetrack_create", "pa_bytetrack_destroy", "pa_bytetrack_reset", "pa_bytetrack_update_batch",
    "pa_model_take_overflow", "pa_yolo_postprocess", "pa_host_register", "pa_host_unregister",
    "pa_engine_bcast_weights_from", "pa_model_fill_arena", "pa_yolo_submit", "pa_yolo_wait",
]


def lib_path() -> Path:
    return _LIB_PATH


def load_library():
    """dlopen libpadel_hip.so (once).  torch is imported first so that the process holds a single
    HIP runtime (torch's bundled libamdhip64.so.7 satisfies our DT_NEEDED by SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise EngineUnavailable(f"{_LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
    if os.environ.get("PADEL_HIP_STANDALONE") != "1":
        import torch  # noqa: F401  (shares one libamdhip64 with RCCL/torch.distributed users)
    lib = C.CDLL(str(_LIB_PATH))
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.pa_abi_version.restype = i32
    lib.pa_device_count.restype = i32
    lib.pa_engine_create.argtypes = [i32, C.POINTER(vp)]
    lib.pa_engine_destroy.argtypes = [vp]
    lib.pa_engine_destroy.restype = None
    lib.pa_last_error.argtypes = [vp]
    lib.pa_last_error.restype = C.c_char_p
    lib.pa_engine_synchronize.argtypes = [vp]
    lib.pa_device_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.pa_device_free.argtypes = [vp, vp]
    lib.pa_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.pa_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    lib.pa_model_create.argtypes = [vp, C.POINTER(pa_model_desc), vp, sz, C.POINTER(vp)]
    lib.pa_model_destroy.argtypes = [vp]
    lib.pa_model_destroy.restype = None
    lib.pa_model_set_max_batch.argtypes = [vp, i32]
    lib.pa_yolo_infer.argtypes = [vp, vp, i32, i32, i32, C.POINTER(pa_yolo_params), vp, vp, vp]
    lib.pa_yolo_submit.argtypes = [vp, vp, i32, i32, i32, C.POINTER(pa_yolo_params), vp, vp, vp, C.POINTER(C.c_int)]
    lib.pa_yolo_wait.argtypes = [vp, i32, C.POINTER(C.c_int)]
    lib.pa_yolo_head_shape.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.pa_yolo_read_head.argtypes = [vp, i32, i32, vp]
    lib.pa_tracknet_infer.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32]
    lib.pa_engine_set_profiling.argtypes = [vp, i32]
    lib.pa_model_last_profile.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.pa_model_profile_text.argtypes = [vp, C.c_char_p, sz]
    lib.pa_ball_create.argtypes = [vp, i32, i32, C.POINTER(vp)]
    lib.pa_ball_destroy.argtypes = [vp]
    lib.pa_ball_destroy.restype = None
    lib.pa_ball_set_background.argtypes = [vp, vp]
    lib.pa_ball_feed.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, C.POINTER(i32)]
    lib.pa_ball_locate.argtypes = [vp, vp, i32, vp]
    lib.pa_ball_background_from_frames.argtypes = [vp, vp, i32, i32, vp]
    lib.pa_upload.argtypes = [vp, vp, vp, sz]
    lib.pa_engine_set_tuning.argtypes = [vp, C.c_char_p, i32]
    lib.pa_engine_set_timeline_path.argtypes = [vp, C.c_char_p]
    lib.pa_model_plan_bytes.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    lib.pa_yolo_netin_shape.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.pa_yolo_read_netin.argtypes = [vp, i32, vp]
    lib.pa_comm_unique_id.argtypes = [vp, sz]
    lib.pa_engine_comm_init.argtypes = [vp, vp, sz, i32, i32]
    lib.pa_engine_comm_destroy.argtypes = [vp]
    lib.pa_engine_comm_destroy.restype = None
    lib.pa_engine_bcast_weights.argtypes = [vp, vp, i32]
    lib.pa_engine_bcast.argtypes = [vp, vp, sz, i32]
    lib.pa_engine_bcast_weights_from.argtypes = [vp, vp, vp, i32]
    lib.pa_engine_allreduce_max.argtypes = [vp, C.POINTER(C.c_double)]
    lib.pa_engine_gather_sizes.argtypes = [vp, sz, C.POINTER(C.c_uint64)]
    lib.pa_engine_gather.argtypes = [vp, vp, sz, vp, sz, C.POINTER(C.c_uint64), i32]
    lib.pa_bytetrack_create.argtypes = [C.c_double, i32, C.c_double, i32, C.POINTER(vp)]
    lib.pa_bytetrack_destroy.argtypes = [vp]
    lib.pa_bytetrack_destroy.restype = None
    lib.pa_bytetrack_reset.argtypes = [vp]
    lib.pa_bytetrack_reset.restype = None
    lib.pa_bytetrack_update_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.pa_model_fill_arena.argtypes = [vp, i32]
    lib.pa_host_register.argtypes = [vp, vp, sz]
    lib.pa_host_unregister.argtypes = [vp, vp]
    lib.pa_model_take_overflow.argtypes = [vp, C.POINTER(i32)]
    lib.pa_yolo_postprocess.argtypes = [vp, C.POINTER(vp), i32, i32, i32, C.POINTER(pa_yolo_params), vp, vp, vp]
    if lib.pa_abi_version() != 5:
        raise EngineUnavailable("libpadel_hip.so ABI version mismatch")
    _lib = lib
    return lib


def graph_dtype(mode: Optional[str] = None) -> str:
    """graph.build_yolov8 / build_tracknet ``dtype`` of the fp32-equivalent path in arithmetic ``mode`` (default: fp32_mode())."""
    return {"h2": "h2", "bx3": "f32"}[mode or fp32_mode()]


class DeviceBuffer:
    """Raw HBM allocation owned by an Engine (bench keeps frame batches resident with it).  ``view(off, n)``
    gives a non-owning window of the same memory (a batch of frames inside a resident clip)."""

    def __init__(self, engine: "Engine", nbytes: int, _ptr: Optional[int] = None):
        self.engine, self.nbytes = engine, nbytes
        self.owner = _ptr is None
        if _ptr is None:
            p = C.c_void_p()
            engine._check(engine.lib.pa_device_malloc(engine.handle, nbytes, C.byref(p)))
            _ptr = p.value
        self.ptr = _ptr

    def view(self, offset: int, nbytes: int) -> "DeviceBuffer":
        assert 0 <= offset and offset + nbytes <= self.nbytes
        return DeviceBuffer(self.engine, nbytes, _ptr=self.ptr + offset)

    def upload(self, arr: np.ndarray, copy_stream: bool = False) -> "DeviceBuffer":
        """copy_stream=True: use the engine's copy stream (pa_upload) — the copy does not queue behind inference
        launched from another host thread."""
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        f = self.engine.lib.pa_upload if copy_stream else self.engine.lib.pa_memcpy_h2d
        self.engine._check(f(self.engine.handle, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, arr: np.ndarray) -> np.ndarray:
        assert arr.flags.c_contiguous and arr.nbytes <= self.nbytes
        self.engine._check(self.engine.lib.pa_memcpy_d2h(self.engine.handle, arr.ctypes.data, self.ptr, arr.nbytes))
        return arr

    def free(self):
        if self.ptr and self.owner:
            self.engine.lib.pa_device_free(self.engine.handle, self.ptr)
        self.ptr = None


class Engine:
    """One engine (HIP stream + device) per GPU."""

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        if self.lib.pa_device_count() <= 0:
            raise EngineUnavailable("no HIP device visible: the padel_analytics_amd engine needs an AMD GPU (gfx950)")
        h = C.c_void_p()
        if self.lib.pa_engine_create(device_id, C.byref(h)) != 0:
            raise EngineUnavailable(self.lib.pa_last_error(None).decode())
        self.handle = h
        self.device_id = device_id

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(self.lib.pa_last_error(self.handle).decode())

    def synchronize(self):
        self._check(self.lib.pa_engine_synchronize(self.handle))

    def set_profiling(self, on: bool):
        self._check(self.lib.pa_engine_set_profiling(self.handle, 1 if on else 0))
        self.profiling = bool(on)

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def pin(self, arr: np.ndarray) -> None:
        """Page-lock a host array in place (hipHostRegister): uploads from it run at PCIe speed.  unpin() before freeing."""
        assert arr.flags.c_contiguous
        self._check(self.lib.pa_host_register(self.handle, arr.ctypes.data, arr.nbytes))

    def unpin(self, arr: np.ndarray) -> None:
        self._check(self.lib.pa_host_unregister(self.handle, arr.ctypes.data))

    def set_tuning(self, **kv):
        """Tests / tools only: impl (2 bx3, 0 tap; 1 — the retired LDS kernel — is refused), variant (tile id, -1 auto), tune, tap_pd, graph, alias, fold_up, timeline."""
        for k, v in kv.items():
            self._check(self.lib.pa_engine_set_tuning(self.handle, k.encode(), int(v)))
            if k == "timeline":
                self.timeline = bool(v)              # (pa_yolo_submit refuses then: yolo.YOLO.submit_frames falls back to infer_frames)

    # ---- multi-GPU: RCCL communicator owned by the library (include/padel_hip.h "multi-GPU")
    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        buf = C.create_string_buffer(bytes(unique_id), len(unique_id))
        self._check(self.lib.pa_engine_comm_init(self.handle, buf, len(unique_id), nranks, rank))
        self.nranks, self.rank = nranks, rank
        if nranks > 1:                     # the sharded runner's gathers travel over this communicator from now on (dist.gather_bytes)
            from . import dist as _dist
            _dist.use_engine_comm(self)

    def bcast_weights(self, model: "Model", root: int = 0):
        self._check(self.lib.pa_engine_bcast_weights(self.handle, model.handle, root))

    def bcast_weights_from(self, src: Optional["Model"], dst: "Model", root: int = 0):
        """Root sends ``src``'s blob, every rank (root included) receives into ``dst`` (created with empty=True)."""
        self._check(self.lib.pa_engine_bcast_weights_from(self.handle, src.handle if src is not None else None, dst.handle, root))

    def bcast(self, buf: DeviceBuffer, nbytes: int, root: int = 0):
        self._check(self.lib.pa_engine_bcast(self.handle, buf.ptr, nbytes, root))

    def gather_bytes(self, buf: np.ndarray, root: int = 0):
        """Variable-length uint8 buffers of every rank -> on `root` the list of them in rank order (None elsewhere), over the
        library's RCCL communicator (``pa_engine_gather_sizes`` + ``pa_engine_gather``: lengths by all-gather, payload by send /
        recv, no padding; torch.distributed is not in this path)."""
        buf = np.ascontiguousarray(buf, np.uint8).reshape(-1)
        n = max(getattr(self, "nranks", 1), 1)
        me = getattr(self, "rank", 0)
        sizes = (C.c_uint64 * n)()
        self._check(self.lib.pa_engine_gather_sizes(self.handle, int(buf.size), sizes))
        total = int(sum(sizes))
        recv = np.empty(max(total, 1), np.uint8) if me == root else None
        self._check(self.lib.pa_engine_gather(self.handle, buf.ctypes.data if buf.size else None, int(buf.size),
                                              recv.ctypes.data if recv is not None else None, total if recv is not None else 0, sizes, root))
        if me != root:
            return None
        out, off = [], 0
        for k in range(n):
            out.append(recv[off:off + int(sizes[k])])
            off += int(sizes[k])
        return out

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(value)
        self._check(self.lib.pa_engine_allreduce_max(self.handle, C.byref(v)))
        return v.value

    def close(self):
        if self.handle:
            from . import dist as _dist
            if _dist._comm_engine is self:
                _dist.use_engine_comm(None)
            self.lib.pa_engine_destroy(self.handle)
            self.handle = None


_default_engines: dict = {}


def comm_unique_id() -> bytes:
    """128-byte RCCL unique id (rank 0 creates it; ship it to the other ranks out of band)."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    if lib.pa_comm_unique_id(buf, 128) != 0:
        raise EngineError(lib.pa_last_error(None).decode())
    return buf.raw


def default_engine(device_id: Optional[int] = None) -> Engine:
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("PADEL_DEVICE") is None \
            else int(os.environ["PADEL_DEVICE"])
        load_library()
        n = _lib.pa_device_count()
        if n > 0:
            device_id %= n
    if device_id not in _default_engines:
        _default_engines[device_id] = Engine(device_id)
    return _default_engines[device_id]


_PAGE = 4096


def _round_up(x: int, a: int) -> int:
    return (int(x) + a - 1) // a * a


class _PinnedBlock:
    """Whole pages of host memory, page-locked for one engine (hipHostRegister locks pages: small arrays that share a page
    cannot be registered / unregistered independently).  Unregistered by ``release()`` or when the object dies — always
    before the memory itself can be freed, which the views handed out keep alive."""

    def __init__(self, engine, nbytes: int):
        import mmap
        size = _round_up(max(int(nbytes), 1), _PAGE)
        # an anonymous mapping of its own, not malloc memory: glibc serves numpy arrays of this size from the brk heap once its
        # dynamic mmap threshold has grown (after the first multi-megabyte array was freed), and pages of the heap that were
        # registered, unregistered and then handed to an unrelated array made a later pageable hipMemcpy from that array fault
        # on the GPU (round 5, tests/test_gpu_pipeline.py::test_close_with_a_ticket_in_flight_then_reuse_the_pages).  A private
        # mapping is page-aligned, zero-filled, and its addresses go back to the kernel — not to malloc — when the views die
        # MAP_PRIVATE | MAP_ANONYMOUS explicitly: Python's default for fileno -1 is MAP_SHARED — shmem-backed pages that a forked child
        # (multiprocessing, the gloo test workers) would share with the parent's result arrays
        self._mm = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE)
        self.block = np.frombuffer(self._mm, np.uint8)
        self.engine = engine
        engine.pin(self.block)
        self._pinned = True

    def release(self) -> None:
        if self._pinned:
            self._pinned = False
            try:
                self.engine.unpin(self.block)
            except Exception as exc:
                # the pages stay registered and their memory is about to be freed: say so (a later copy from memory that
                # lands on these addresses fails inside the runtime) — except at interpreter shutdown, when the engine is gone
                if _sys is not None and getattr(_sys, "meta_path", None) is not None and getattr(self.engine, "handle", None):
                    print(f"padel_analytics_amd: hipHostUnregister of a result block failed ({exc}); pages left registered", file=_sys.stderr)

    def __del__(self):
        self.release()


class Model:
    """A graph + weights resident in HBM on one engine."""

    def __init__(self, engine: Engine, graph: G.Graph, blob: Optional[np.ndarray] = None, *, empty: bool = False):
        """empty=True: allocate the weight blob in HBM without uploading anything — it arrives through
        ``engine.bcast_weights(model)`` from the rank that loaded the checkpoint."""
        self.engine, self.graph = engine, graph
        if not empty:
            blob = graph.blob() if blob is None else np.ascontiguousarray(blob, np.float32)
        bufs = (pa_buf_desc * len(graph.bufs))(*[pa_buf_desc(l, c) for l, c in graph.bufs])
        ops = (pa_op_desc * len(graph.ops))()
        for i, o in enumerate(graph.ops):
            for k, v in o.items():
                setattr(ops[i], k, int(v))
        d = pa_model_desc(task=graph.task, nc=graph.nc, nk=graph.nk, kpt_dim=graph.kpt_dim, n_bufs=len(graph.bufs),
                          bufs=bufs, n_ops=len(graph.ops), ops=ops, in_channels=graph.in_channels,
                          dtype=getattr(graph, "dtype", 0))
        for i in range(3):
            d.head_buf[i] = graph.head_buf[i] if i < len(graph.head_buf) else -1
        h = C.c_void_p()
        engine._check(engine.lib.pa_model_create(engine.handle, C.byref(d), None if empty else blob.ctypes.data,
                                                 graph.n_floats if empty else blob.size, C.byref(h)))
        self.handle = h
        self.max_batch = 64
        self._out_ring: dict = {}        # (n, max_det) -> [three page-locked (boxes, kpts, counts) sets, next slot]

    #: tickets a model may have in flight (PA_MAX_INFLIGHT of include/padel_hip.h)
    MAX_INFLIGHT = 4
    #: recycled result sets per (n, max_det): every ticket in flight owns one, plus the set whose results the host stage is
    #: still reading and the one collected just before it — a set is handed out again OUT_RING calls later
    OUT_RING = MAX_INFLIGHT + 2

    def _ring_outputs(self, n: int, max_det: int):
        key = (int(n), int(max_det))
        ring = self._out_ring.get(key)
        if ring is None:
            nk = self.graph.nk
            sizes = [n * max_det * 6 * 4, n * max_det * nk * 4, n * 4]
            offs = [0, _round_up(sizes[0], 64), 0]
            offs[2] = offs[1] + _round_up(sizes[1], 64)
            sets = []
            for _ in range(self.OUT_RING):
                pb = _PinnedBlock(self.engine, offs[2] + sizes[2])
                block = pb.block
                boxes = block[offs[0]:offs[0] + sizes[0]].view(np.float32).reshape(n, max_det, 6)
                kpts = block[offs[1]:offs[1] + sizes[1]].view(np.float32).reshape(n, max_det, nk) if nk else None
                counts = block[offs[2]:offs[2] + sizes[2]].view(np.int32).reshape(n)
                sets.append((boxes, kpts, counts, pb))
            ring = self._out_ring[key] = [sets, 0]
        sets, i = ring
        ring[1] = (i + 1) % self.OUT_RING
        return sets[i][:3]

    def _free_rings(self):
        for sets, _ in self._out_ring.values():
            for arrs in sets:
                arrs[3].release()
        self._out_ring = {}

    def __del__(self):
        # a model dropped without close(): its page-locked result blocks must be unregistered BEFORE their memory goes back
        # to the allocator (a stale registration makes later copies to / from whatever reuses those pages fail) — and AFTER
        # the stream has drained: a ticket still queued copies its results into exactly those pages
        # (only that: destroying the HBM side here could outlive its engine)
        try:
            if self._out_ring and getattr(self, "handle", None) and self.engine.handle:
                self.engine.synchronize()
        except Exception:
            pass
        try:
            self._free_rings()
        except Exception:
            pass

    def set_max_batch(self, n: int):
        self.engine._check(self.engine.lib.pa_model_set_max_batch(self.handle, int(n)))
        self.max_batch = int(n)

    def yolo_infer(self, frames, n: int, h: int, w: int, *, imgsz: int, conf: float, iou: float,
                   classes: Optional[Sequence[int]] = None, max_det: int = 300, pre_mode: int = PRE_LETTERBOX,
                   channel_reverse: bool = False, letterbox_auto: bool = True, reuse_outputs: bool = False):
        """frames: (n,h,w,3) uint8 ndarray, or a DeviceBuffer holding the same bytes.
        Returns (boxes (n,max_det,6), kpts (n,max_det,nk) | None, counts (n,)).

        ``reuse_outputs=True`` (the trackers' batch loops): the arrays are one of ``OUT_RING`` page-locked sets this model
        keeps per (n, max_det) — the device copies its results straight into them (no staging copy, no 3.5 MB of fresh
        zeros per pose batch) — and are overwritten by the ``OUT_RING``-th next such call; copy what must live longer."""
        on_dev = isinstance(frames, DeviceBuffer)
        if on_dev:
            ptr = frames.ptr
            assert frames.nbytes >= n * h * w * 3
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            assert frames.shape == (n, h, w, 3), frames.shape
            ptr = frames.ctypes.data
        cls_arr = None
        p = pa_yolo_params(imgsz=imgsz, pre_mode=pre_mode, channel_reverse=int(channel_reverse),
                           letterbox_auto=int(letterbox_auto), conf=conf, iou=iou, max_det=max_det,
                           n_classes=0, classes=None, frames_on_device=int(on_dev))
        if classes is not None and len(classes):
            cls_arr = (C.c_int32 * len(classes))(*[int(c) for c in classes])
            p.n_classes = len(classes)
            p.classes = cls_arr
        if reuse_outputs:
            boxes, kpts, counts = self._ring_outputs(n, max_det)
        else:
            boxes = np.zeros((n, max_det, 6), np.float32)
            counts = np.zeros((n,), np.int32)
            nk = self.graph.nk
            kpts = np.zeros((n, max_det, nk), np.float32) if nk else None
        self.engine._check(self.engine.lib.pa_yolo_infer(
            self.handle, ptr, n, h, w, C.byref(p), boxes.ctypes.data,
            kpts.ctypes.data if kpts is not None else None, counts.ctypes.data))
        return boxes, kpts, counts

    def yolo_submit(self, frames: "DeviceBuffer", n: int, h: int, w: int, *, imgsz: int, conf: float, iou: float,
                    classes: Optional[Sequence[int]] = None, max_det: int = 300, pre_mode: int = PRE_LETTERBOX,
                    channel_reverse: bool = False, letterbox_auto: bool = True):
        """``yolo_infer`` in two halves (pa_yolo_submit / pa_yolo_wait): enqueue the whole call behind what the engine's stream
        still holds and return a ticket at once; ``yolo_wait(ticket)`` -> (boxes, kpts, counts, overflow).  Frames must be in
        HBM; the results land in one of the model's recycled page-locked sets (``reuse_outputs`` rules).  Submit batch k + 1
        before waiting for batch k and the GPU never waits for the host between batches."""
        assert isinstance(frames, DeviceBuffer) and frames.nbytes >= n * h * w * 3
        cls_arr = None
        p = pa_yolo_params(imgsz=imgsz, pre_mode=pre_mode, channel_reverse=int(channel_reverse),
                           letterbox_auto=int(letterbox_auto), conf=conf, iou=iou, max_det=max_det,
                           n_classes=0, classes=None, frames_on_device=1)
        if classes is not None and len(classes):
            cls_arr = (C.c_int32 * len(classes))(*[int(c) for c in classes])
            p.n_classes = len(classes)
            p.classes = cls_arr
        boxes, kpts, counts = self._ring_outputs(n, max_det)
        t = C.c_int(-1)
        self.engine._check(self.engine.lib.pa_yolo_submit(
            self.handle, frames.ptr, n, h, w, C.byref(p), boxes.ctypes.data,
            kpts.ctypes.data if kpts is not None else None, counts.ctypes.data, C.byref(t)))
        return (t.value, boxes, kpts, counts)

    def yolo_wait(self, ticket):
        t, boxes, kpts, counts = ticket
        ovf = C.c_int(0)
        self.engine._check(self.engine.lib.pa_yolo_wait(self.handle, t, C.byref(ovf)))
        return boxes, kpts, counts, bool(ovf.value)

    def head_shapes(self, h: int, w: int, imgsz: int, pre_mode: int = PRE_LETTERBOX) -> list:
        """[(H_l, W_l, c)] of the three head maps for source size h x w (host arithmetic of the planner)."""
        if pre_mode == PRE_PIL_STRETCH:
            nh = nw = imgsz
        else:
            r = min(imgsz / h, imgsz / w)
            rw, rh = int(round(w * r)), int(round(h * r))
            nw, nh = rw + (imgsz - rw) % 32, rh + (imgsz - rh) % 32
        c = self.graph.bufs[self.graph.head_buf[0]][1]
        return [(nh >> (3 + l), nw >> (3 + l), c) for l in range(3)]

    def yolo_postprocess(self, heads, h: int, w: int, *, imgsz: int, conf: float, iou: float,
                         classes: Optional[Sequence[int]] = None, max_det: int = 300, pre_mode: int = PRE_LETTERBOX):
        """Decode + NMS + rescale of caller-supplied head maps (``heads[l]``: (n, H_l, W_l, c) fp32, ``head_shapes``):
        the Detect / Pose inference branch alone — tests feed hand-derived cases through it."""
        heads = [np.ascontiguousarray(x, np.float32) for x in heads]
        n = heads[0].shape[0]
        assert [x.shape[1:] for x in heads] == [tuple(s) for s in self.head_shapes(h, w, imgsz, pre_mode)], [x.shape for x in heads]
        p = pa_yolo_params(imgsz=imgsz, pre_mode=pre_mode, channel_reverse=0, letterbox_auto=1, conf=conf, iou=iou,
                           max_det=max_det, n_classes=0, classes=None, frames_on_device=0)
        cls_arr = None
        if classes is not None and len(classes):
            cls_arr = (C.c_int32 * len(classes))(*[int(c) for c in classes])
            p.n_classes = len(classes)
            p.classes = cls_arr
        ptrs = (C.c_void_p * 3)(*[x.ctypes.data for x in heads])
        boxes = np.zeros((n, max_det, 6), np.float32)
        counts = np.zeros((n,), np.int32)
        nk = self.graph.nk
        kpts = np.zeros((n, max_det, nk), np.float32) if nk else None
        self.engine._check(self.engine.lib.pa_yolo_postprocess(self.handle, ptrs, n, h, w, C.byref(p), boxes.ctypes.data,
                                                               kpts.ctypes.data if kpts is not None else None, counts.ctypes.data))
        return boxes, kpts, counts

    def fill_arena(self, byte_value: int = 0xFF) -> None:
        """Tests: overwrite the planned activation arena (0xFF = NaN patterns): a following inference must not notice."""
        self.engine._check(self.engine.lib.pa_model_fill_arena(self.handle, int(byte_value)))

    def take_overflow(self) -> bool:
        """h2 models: True if an activation written since the last call did not fit the fp16 range (clears the flag)."""
        v = C.c_int(0)
        self.engine._check(self.engine.lib.pa_model_take_overflow(self.handle, C.byref(v)))
        return bool(v.value)

    def read_head(self, level: int, n: int) -> np.ndarray:
        hh, ww, cc = C.c_int(), C.c_int(), C.c_int()
        self.engine._check(self.engine.lib.pa_yolo_head_shape(self.handle, level, C.byref(hh), C.byref(ww), C.byref(cc)))
        out = np.empty((n, hh.value, ww.value, cc.value), np.float32)
        self.engine._check(self.engine.lib.pa_yolo_read_head(self.handle, level, n, out.ctypes.data))
        return out

    def read_netin(self, n: int) -> np.ndarray:
        """u8 NHWC4 network input of the first n frames of the last yolo_infer call."""
        hh, ww = C.c_int(), C.c_int()
        self.engine._check(self.engine.lib.pa_yolo_netin_shape(self.handle, C.byref(hh), C.byref(ww)))
        out = np.empty((n, hh.value, ww.value, 4), np.uint8)
        self.engine._check(self.engine.lib.pa_yolo_read_netin(self.handle, n, out.ctypes.data))
        return out

    def plan_bytes(self) -> tuple:
        """(arena bytes allocated, sum of the logical buffers) of the current activation plan."""
        a, l = C.c_size_t(), C.c_size_t()
        self.engine._check(self.engine.lib.pa_model_plan_bytes(self.handle, C.byref(a), C.byref(l)))
        return a.value, l.value

    def tracknet_infer(self, x: np.ndarray) -> np.ndarray:
        """x: (n, H, W, C_in) NHWC (fp32; fp16 for a graph with dtype f16) -> (n, H, W, C_out) fp32."""
        x = np.ascontiguousarray(x, np.float16 if getattr(self.graph, "dtype", 0) == G.DTYPE_F16 else np.float32)
        n, h, w, c = x.shape
        assert c == self.graph.bufs[0][1], (c, self.graph.bufs[0])
        lvl, cout = self.graph.bufs[self.graph.head_buf[0]]
        out = np.empty((n, h >> lvl, w >> lvl, cout), np.float32)
        self.engine._check(self.engine.lib.pa_tracknet_infer(self.handle, x.ctypes.data, n, h, w, 0, out.ctypes.data, 0))
        return out

    def last_profile(self, cap: int = 4096):
        kinds = np.zeros(cap, np.int32)
        ms = np.zeros(cap, np.float32)
        fl = np.zeros(cap, np.float64)
        ks = np.zeros(cap, np.int32)
        n = self.engine.lib.pa_model_last_profile(self.handle, cap, kinds.ctypes.data, ms.ctypes.data, fl.ctypes.data,
                                                  ks.ctypes.data)
        return [dict(kind=int(kinds[i]), ms=float(ms[i]), flops=float(fl[i]), ksize=int(ks[i])) for i in range(n)]

    def profile_rows(self):
        """Per-op records of the last profiled inference: dicts with shape, tile and milliseconds."""
        buf = C.create_string_buffer(1 << 20)
        self.engine.lib.pa_model_profile_text(self.handle, buf, len(buf))
        rows = []
        for line in buf.value.decode().splitlines():
            k, ks, M, co, ci, st, mf, nf, ms, fl, res = (line.split(",") + ["0"])[:11]
            rows.append(dict(kind=int(k), ksize=int(ks), M=int(M), cout=int(co), cin=int(ci), stride=int(st),
                             mf=int(mf), nf=int(nf), ms=float(ms), flops=float(fl), res=int(res)))
        return rows

    def close(self):
        """Drain first, unpin second: pa_model_destroy synchronizes the engine's stream, so tickets still queued (the
        overflow fallback closes a model with batch k + 1 in flight) have finished their copies into the page-locked result
        sets before those pages are unregistered (tests/test_gpu_pipeline.py closes a model with a ticket out)."""
        if self.handle:
            self.engine.lib.pa_model_destroy(self.handle)
            self.handle = None
            self._free_rings()


class NativeByteTrack:
    """``pa_bytetrack_*``: the host-native (C++) twin of ``padel_analytics_amd.bytetrack.ByteTrack`` — same
    algorithm, orderings and id semantics (tests/test_bytetrack_golden.py pins both to the same fixtures); consumes
    a whole batch of frames per call.  Host code inside libpadel_hip.so: works without a GPU."""

    def __init__(self, track_activation_threshold: float = 0.25, lost_track_buffer: int = 30,
                 minimum_matching_threshold: float = 0.8, frame_rate: int = 30):
        self.lib = load_library()
        h = C.c_void_p()
        if self.lib.pa_bytetrack_create(track_activation_threshold, lost_track_buffer, minimum_matching_threshold,
                                        int(frame_rate), C.byref(h)) != 0:
            raise EngineError("pa_bytetrack_create failed")
        self.handle = h

    def reset(self) -> None:
        self.lib.pa_bytetrack_reset(self.handle)

    def update_batch(self, boxes: np.ndarray, counts: np.ndarray, keep: Optional[np.ndarray] = None) -> np.ndarray:
        """boxes (n, stride, 6) fp32, counts (n,) int32, keep (n, stride) bool/uint8 | None -> ids (n, stride) int32."""
        boxes = np.ascontiguousarray(boxes, np.float32)
        counts = np.ascontiguousarray(counts, np.int32)
        n, stride, six = boxes.shape
        assert six == 6 and counts.shape == (n,)
        kp = None
        if keep is not None:
            keep = np.ascontiguousarray(keep, np.uint8)
            assert keep.shape == (n, stride)
            kp = keep.ctypes.data
        ids = np.empty((n, stride), np.int32)
        if self.lib.pa_bytetrack_update_batch(self.handle, boxes.ctypes.data, counts.ctypes.data, kp, n, stride,
                                              ids.ctypes.data) != 0:
            raise EngineError("pa_bytetrack_update_batch failed")
        return ids

    def update_with_detections(self, detections):
        """Single-frame form with the signature of ``sv.ByteTrack.update_with_detections`` (players_tracker.py:367)."""
        n = len(detections)
        boxes = np.zeros((1, max(n, 1), 6), np.float32)
        boxes[0, :n, :4] = detections.xyxy
        boxes[0, :n, 4] = 1.0 if detections.confidence is None else detections.confidence
        ids = self.update_batch(boxes, np.array([n], np.int32))[0, :n]
        out = detections[np.nonzero(ids >= 0)[0]]
        out.tracker_id = ids[ids >= 0].astype(int)
        return out

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.pa_bytetrack_destroy(self.handle)
            self.handle = None


class BallSession:
    """Streaming TrackNet session on one engine (``pa_ball_*``): resize + windows + network + temporal
    ensemble + threshold on the GPU; yields one 288x512 uint8 mask (and optionally the fp32 heat map) per frame."""

    H, W = 288, 512

    def __init__(self, model: Model, src_h: int, src_w: int):
        self.model, self.src_h, self.src_w = model, src_h, src_w
        h = C.c_void_p()
        model.engine._check(model.engine.lib.pa_ball_create(model.handle, src_h, src_w, C.byref(h)))
        self.handle = h
        self.max_feed = model.max_batch

    def set_background(self, median_rgb: np.ndarray):
        m = np.ascontiguousarray(median_rgb, np.uint8)
        assert m.shape == (self.src_h, self.src_w, 3), m.shape
        self.model.engine._check(self.model.engine.lib.pa_ball_set_background(self.handle, m.ctypes.data))

    def background_from_frames(self, frames_bgr, want_median: bool = False, n: Optional[int] = None):
        """Median background of (n, h, w, 3) uint8 BGR frames on the device (np.median + uint8 truncation).
        ``frames_bgr``: ndarray, or a DeviceBuffer holding ``n`` frames."""
        on_dev = isinstance(frames_bgr, DeviceBuffer)
        if on_dev:
            ptr = frames_bgr.ptr
            assert n is not None and frames_bgr.nbytes >= n * self.src_h * self.src_w * 3
        else:
            f = np.ascontiguousarray(frames_bgr, np.uint8)
            n = len(f)
            assert f.shape == (n, self.src_h, self.src_w, 3)
            ptr = f.ctypes.data
        med = np.empty((self.src_h, self.src_w, 3), np.uint8) if want_median else None
        self.model.engine._check(self.model.engine.lib.pa_ball_background_from_frames(
            self.handle, ptr, n, int(on_dev), med.ctypes.data if want_median else None))
        return med

    def feed(self, frames_bgr, flush: bool = False, want_heat: bool = False,
             want_rects: bool = False, want_masks: bool = True, n: Optional[int] = None):
        """frames_bgr: (n, h, w, 3) uint8 with n <= max_feed, a DeviceBuffer holding ``n`` such frames, or None with
        flush=True.  Returns (masks (k,288,512) uint8 | None, heat (k,288,512) fp32 | None[, rects (k,4) int32]):
        outputs for the next k frames in order; rects = predict_location of each mask computed on the device."""
        on_dev = isinstance(frames_bgr, DeviceBuffer)
        ptr = None
        if on_dev:
            assert n is not None and 0 < n <= self.max_feed and frames_bgr.nbytes >= n * self.src_h * self.src_w * 3
            ptr = frames_bgr.ptr
        else:
            n = 0 if frames_bgr is None else len(frames_bgr)
            if n:
                frames_bgr = np.ascontiguousarray(frames_bgr, np.uint8)
                assert frames_bgr.shape == (n, self.src_h, self.src_w, 3) and n <= self.max_feed
                ptr = frames_bgr.ctypes.data
        masks = np.empty((n + 7, self.H, self.W), np.uint8) if (want_masks or not want_rects) else None
        heat = np.empty((n + 7, self.H, self.W), np.float32) if want_heat else None
        rects = np.empty((n + 7, 4), np.int32) if want_rects else None
        cnt = C.c_int(0)
        self.model.engine._check(self.model.engine.lib.pa_ball_feed(
            self.handle, ptr, n, int(on_dev), int(flush), masks.ctypes.data if masks is not None else None,
            heat.ctypes.data if want_heat else None, rects.ctypes.data if want_rects else None, C.byref(cnt)))
        k = cnt.value
        out = (None if masks is None else masks[:k], heat[:k] if want_heat else None)
        return out + (rects[:k],) if want_rects else out

    def locate(self, masks: np.ndarray) -> np.ndarray:
        """predict_location of (n,288,512) uint8 masks on the device -> (n,4) int32 {x, y, w, h}."""
        masks = np.ascontiguousarray(masks, np.uint8)
        n = len(masks)
        assert masks.shape == (n, self.H, self.W)
        rects = np.empty((n, 4), np.int32)
        self.model.engine._check(self.model.engine.lib.pa_ball_locate(self.handle, masks.ctypes.data, n, rects.ctypes.data))
        return rects

    def close(self):
        if self.handle:
            self.model.engine.lib.pa_ball_destroy(self.handle)
            self.handle = None
