"""Multi-GPU plumbing: frames are independent, so the path shards by batch with no
data-path collective; the only exchange is a ONE-TIME broadcast of the packed weight blob from the rank
that loaded the checkpoint (RCCL over xGMI on the GPU box — ``torch.distributed`` backend "nccl"; "gloo" in
the CPU tests).  ByteTrack ids are assigned on rank 0 after the gather because the tracker is sequential in
global frame order."""
from __future__ import annotations

from typing import Optional

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple:
    """Contiguous block of frame indices owned by `rank` (contiguous so results concatenate in frame order)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_array(arr: Optional[np.ndarray], shape: tuple, dtype, src: int = 0) -> np.ndarray:
    """Small host array (the ball tracker's background median) from `src` to every rank."""
    import torch
    import torch.distributed as dist
    if world_size() == 1:
        assert arr is not None
        return arr
    if rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype).reshape(shape).copy())
    else:
        t = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def broadcast_flag(value: bool, src: int = 0) -> bool:
    """A decision taken on `src` (e.g. "this tracker's predictions are already cached") that every rank must follow
    before entering a collective path."""
    import torch.distributed as dist
    if world_size() == 1:
        return bool(value)
    box = [bool(value) if rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return bool(box[0])


def any_flag(value: bool) -> bool:
    """True on every rank iff `value` is true on at least one (a collective: every rank must call it)."""
    import torch
    import torch.distributed as dist
    if world_size() == 1:
        return bool(value)
    t = torch.tensor([1 if value else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.cpu()[0]))


def share_unique_id(make_id) -> bytes:
    """RCCL bootstrap for ``Engine.comm_init``: rank 0 calls ``make_id()`` (``engine.comm_unique_id``), the 128
    bytes reach the other ranks through the process group's store (TCP, out of band of the data path)."""
    import torch.distributed as dist
    if world_size() == 1:
        return make_id()
    box = [make_id() if rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def broadcast_blob(blob: Optional[np.ndarray], n_floats: int, src: int = 0, device=None) -> np.ndarray:
    """Broadcast the fp32 weight blob; ranks != src pass None.  `device`: torch device the collective runs on
    ("cuda:k" for RCCL, None/cpu for gloo)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert blob is not None
        return blob
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(blob, np.float32))
    else:
        t = torch.empty(n_floats, dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def _pack_arrays(arrays: list) -> np.ndarray:
    """[ndarray] -> one uint8 buffer: 8-byte header length, pickled [(dtype, shape)], then the raw bytes back to back."""
    import pickle
    arrays = [np.ascontiguousarray(a) for a in arrays]
    meta = pickle.dumps([(a.dtype.str, a.shape) for a in arrays], protocol=4)
    out = np.empty(8 + len(meta) + sum(a.nbytes for a in arrays), np.uint8)
    out[:8] = np.frombuffer(np.int64(len(meta)).tobytes(), np.uint8)
    out[8:8 + len(meta)] = np.frombuffer(meta, np.uint8)
    pos = 8 + len(meta)
    for a in arrays:
        out[pos:pos + a.nbytes] = a.reshape(-1).view(np.uint8)
        pos += a.nbytes
    return out


def _unpack_arrays(buf: np.ndarray) -> list:
    import pickle
    nmeta = int(np.frombuffer(buf[:8].tobytes(), np.int64)[0])
    meta = pickle.loads(buf[8:8 + nmeta].tobytes())
    pos, out = 8 + nmeta, []
    for dt, shape in meta:
        nb = int(np.dtype(dt).itemsize * int(np.prod(shape, dtype=np.int64)))
        out.append(np.frombuffer(buf[pos:pos + nb].tobytes(), np.dtype(dt)).reshape(shape))
        pos += nb
    return out


_comm_engine = None       # the Engine whose library-owned RCCL communicator carries the gathers (set by Engine.comm_init)


def use_engine_comm(engine) -> None:
    """Round 6: route ``gather_bytes`` (and with it gather_arrays / gather_results) through the communicator ``libpadel_hip.so`` owns
    (``pa_engine_gather``: ncclAllGather of the lengths + ncclSend / ncclRecv, no padding) instead of torch.distributed.  Called by
    ``Engine.comm_init`` for nranks > 1; ``None`` switches back (CPU tests: gloo)."""
    global _comm_engine
    _comm_engine = engine


def gather_bytes(buf: np.ndarray, dst: int = 0) -> Optional[list]:
    """One uint8 buffer per rank -> on `dst` the list of all ranks' buffers (rank order), None elsewhere.  On the GPU box: the
    library's own RCCL communicator (``use_engine_comm``).  Without one (the gloo CPU tests): torch.distributed."""
    eng = _comm_engine
    if eng is not None and getattr(eng, "nranks", 1) == world_size() and world_size() > 1:
        return eng.gather_bytes(np.ascontiguousarray(buf, np.uint8).reshape(-1), dst)
    return _gather_bytes_torch(buf, dst)


def _gather_bytes_torch(buf: np.ndarray, dst: int = 0) -> Optional[list]:
    """The torch.distributed form (gloo in the CPU tests): an all-gather of the lengths and ONE gather of the buffers padded to
    the longest."""
    import torch
    import torch.distributed as dist
    world, me = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    dev = (lambda t: t.cuda()) if on_gpu else (lambda t: t)
    n = dev(torch.tensor([int(buf.size)], dtype=torch.int64))
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.cpu()[0]) for x in sizes]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=torch.uint8)
    send[:buf.size] = torch.from_numpy(np.ascontiguousarray(buf, np.uint8).reshape(-1))
    send = dev(send)
    recv = [dev(torch.empty(cap, dtype=torch.uint8)) for _ in range(world)] if me == dst else None
    dist.gather(send, recv, dst=dst)
    if me != dst:
        return None
    return [recv[r].cpu().numpy()[:sizes[r]] for r in range(world)]


def gather_arrays(arrays: list, dst: int = 0) -> Optional[list]:
    """Every rank contributes a list of ndarrays; `dst` gets [rank 0's list, rank 1's list, ...] — O(bytes): no per-item
    pickling (the sharded runner's partials: counts + rows per rank, ``Tracker.pack_partials``)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(arrays)]
    got = gather_bytes(_pack_arrays(arrays), dst)
    return None if got is None else [_unpack_arrays(b) for b in got]


def gather_results(local: list, dst: int = 0) -> Optional[list]:
    """Gather per-rank python result lists (already in local frame order) to `dst`, concatenated in rank
    (= global frame) order.  Fallback for partials a tracker cannot express as arrays (``Tracker.pack_partials`` returns
    None): the list is pickled once per rank and travels as ONE byte buffer."""
    import pickle
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    got = gather_bytes(np.frombuffer(pickle.dumps(local, protocol=4), np.uint8), dst)
    if got is None:
        return None
    return [x for b in got for x in pickle.loads(b.tobytes())]
