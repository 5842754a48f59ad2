"""Device plumbing: torch is used for HBM allocation, streams and torch.distributed only."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib


def require_gpu() -> torch.device:
    """The device this process drives (one process per GPU); raises when there is none."""
    if not torch.cuda.is_available():
        raise _lib.ScamdError(
            "scanpy_amd needs an AMD GPU (MI355X/gfx950): torch.cuda.is_available() is False "
            "and there is deliberately no CPU fallback"
        )
    _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def set_device_from_env() -> torch.device:
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    return torch.device("cuda", local_rank)


def ptr(t: torch.Tensor | None) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "C-ABI expects contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _WorkspacePool:
    """One grow-only scratch buffer per device, reused across calls (caller-owned workspace)."""

    def __init__(self) -> None:
        self._buf: dict[int, torch.Tensor] = {}

    def get(self, nbytes: int, device: torch.device) -> torch.Tensor:
        key = device.index or 0
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            self._buf.pop(key, None)
            buf = None
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf

    def release(self) -> None:
        self._buf.clear()


workspace_pool = _WorkspacePool()


class _PinnedUploader:
    """Host array -> HBM at PCIe rate.  `tensor.to(device)` of a PAGEABLE numpy array goes through the runtime's own
    bounce buffer, one memcpy thread deep: 0.8 GB of CSR took 30 ms warm (27 GB/s), more than the whole PCA that waits
    for it.  Here the array is cut into 32 MB pieces; a small thread pool copies piece i + 1 into one of two page-locked
    staging buffers (numpy releases the GIL for the copy) while piece i is DMA'd out of the other on a copy stream.
    The staging buffers are allocated once per process (page-locking costs more than the copy)."""

    PIECE = int(os.environ.get("SCAMD_UPLOAD_PIECE_MB", "32")) << 20
    THREADS = int(os.environ.get("SCAMD_UPLOAD_THREADS", "8"))

    def __init__(self) -> None:
        self._stage = None
        self._events = None
        self._pool = None
        self._stream = None

    def _setup(self, device: torch.device) -> None:
        from concurrent.futures import ThreadPoolExecutor

        self._stage = [torch.empty(self.PIECE, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        self._events = [torch.cuda.Event() for _ in range(2)]
        self._pool = ThreadPoolExecutor(max_workers=self.THREADS)
        self._stream = torch.cuda.Stream(device=device)

    def upload(self, arr, device: torch.device) -> torch.Tensor:
        import numpy as np

        arr = np.ascontiguousarray(arr)
        src = torch.from_numpy(arr)
        nbytes = arr.nbytes
        # (small arrays, and arrays that already live in page-locked memory -- e.g. the slots `to_host` produced: direct DMA)
        if nbytes < (8 << 20) or os.environ.get("SCAMD_PINNED_UPLOAD") == "0" or src.is_pinned():
            return src.to(device)
        if self._stage is None:
            self._setup(device)
        out = torch.empty(arr.shape, dtype=src.dtype, device=device)
        out_b = out.view(torch.uint8).reshape(-1)
        src_b = arr.view(np.uint8).reshape(-1)
        cur = torch.cuda.current_stream(device)
        self._stream.wait_stream(cur)  # `out` was allocated on the current stream
        n_piece = (nbytes + self.PIECE - 1) // self.PIECE
        for i in range(n_piece):
            lo, hi = i * self.PIECE, min(nbytes, (i + 1) * self.PIECE)
            slot = i & 1
            stage = self._stage[slot]
            self._events[slot].synchronize()  # the last DMA out of this staging buffer (this call's or an earlier one's)
            dst = stage.numpy()
            step = (hi - lo + self.THREADS - 1) // self.THREADS
            futs = [self._pool.submit(np.copyto, dst[a - lo:min(a + step, hi) - lo], src_b[a:min(a + step, hi)])
                    for a in range(lo, hi, step)]
            for f in futs:
                f.result()
            with torch.cuda.stream(self._stream):
                out_b[lo:hi].copy_(stage[:hi - lo], non_blocking=True)
                self._events[slot].record(self._stream)
        cur.wait_stream(self._stream)
        out.record_stream(self._stream)  # written on the copy stream, allocated (and later freed) on the current one
        return out


pinned_uploader = _PinnedUploader()


def to_host(t: torch.Tensor):
    """Device tensor -> numpy array.  Large results (the 120 MB of kNN distances, the 200 MB of connectivities at 1M
    cells) land in PAGE-LOCKED host memory: `t.cpu()` into pageable memory goes through the runtime's bounce buffer at
    less than half the PCIe rate.  The page-locked block comes from torch's caching host allocator (allocated once per
    size, recycled when the array is released); the numpy array keeps it alive.  SCAMD_PINNED_DOWNLOAD=0: plain `.cpu()`."""
    if not t.is_cuda or t.numel() * t.element_size() < (8 << 20) or os.environ.get("SCAMD_PINNED_DOWNLOAD") == "0":
        return t.cpu().numpy()
    t = t.contiguous()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()
