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


class _PinnedTransfer:
    """Host array <-> HBM at PCIe rate.  `tensor.to(device)` of a PAGEABLE numpy array goes through the runtime's own
    bounce buffer, one memcpy thread deep: 0.8 GB of CSR took 30 ms warm (27 GB/s), more than the whole PCA that waits
    for it.  Here the array is cut into 32 MB pieces; a small thread pool copies piece i + 1 into one of two page-locked
    staging buffers (numpy releases the GIL for the copy) while piece i is DMA'd out of the other on a copy stream;
    downloads run the same pipeline backwards, so the arrays handed to the user (`.obsm` / `.obsp` slots) live in ordinary
    pageable memory and nothing stays page-locked beyond the two staging buffers.
    Staging buffers, events and the copy stream are kept PER DEVICE (a stream belongs to one device: a singleton created
    for the first device would record its events on the wrong stream for the second); one transfer at a time (lock)."""

    PIECE = int(os.environ.get("SCAMD_UPLOAD_PIECE_MB", "32")) << 20
    THREADS = int(os.environ.get("SCAMD_UPLOAD_THREADS", "8"))

    def __init__(self) -> None:
        import threading

        self._per_device: dict[int, tuple] = {}
        self._pool = None
        self._lock = threading.Lock()

    def _state(self, device: torch.device):
        from concurrent.futures import ThreadPoolExecutor

        key = device.index if device.index is not None else torch.cuda.current_device()
        st = self._per_device.get(key)
        if st is None:
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=self.THREADS)
            dev = torch.device("cuda", key)
            st = ([torch.empty(self.PIECE, dtype=torch.uint8, pin_memory=True) for _ in range(2)],
                  [torch.cuda.Event() for _ in range(2)], torch.cuda.Stream(device=dev))
            self._per_device[key] = st
        return st

    def _host_copy(self, dst, src) -> None:
        """dst[:] = src for two flat uint8 numpy views of equal length, split over the thread pool"""
        import numpy as np

        n = dst.shape[0]
        step = (n + self.THREADS - 1) // self.THREADS
        futs = [self._pool.submit(np.copyto, dst[a:min(a + step, n)], src[a:min(a + step, n)]) for a in range(0, n, step)]
        for f in futs:
            f.result()

    def upload(self, arr, device: torch.device) -> torch.Tensor:
        import numpy as np

        arr = np.ascontiguousarray(arr)
        src = torch.from_numpy(arr)
        nbytes = arr.nbytes
        # (small arrays, and arrays that already live in page-locked memory: direct DMA)
        if nbytes < (8 << 20) or os.environ.get("SCAMD_PINNED_UPLOAD") == "0" or src.is_pinned():
            return src.to(device)
        with self._lock:
            stages, events, stream = self._state(device)
            out = torch.empty(arr.shape, dtype=src.dtype, device=device)
            out_b = out.view(torch.uint8).reshape(-1)
            src_b = arr.view(np.uint8).reshape(-1)
            cur = torch.cuda.current_stream(device)
            stream.wait_stream(cur)  # `out` was allocated on the current stream
            n_piece = (nbytes + self.PIECE - 1) // self.PIECE
            for i in range(n_piece):
                lo, hi = i * self.PIECE, min(nbytes, (i + 1) * self.PIECE)
                slot = i & 1
                stage = stages[slot]
                events[slot].synchronize()  # the last DMA out of this staging buffer (this call's or an earlier one's)
                self._host_copy(stage.numpy()[:hi - lo], src_b[lo:hi])
                with torch.cuda.stream(stream):
                    out_b[lo:hi].copy_(stage[:hi - lo], non_blocking=True)
                    events[slot].record(stream)
            cur.wait_stream(stream)
            out.record_stream(stream)  # written on the copy stream, allocated (and later freed) on the current one
            return out

    def upload_into(self, arr, out: torch.Tensor) -> None:
        """`upload` into a device tensor the caller allocated (same dtype and element count): the pieces are ordered behind
        the CURRENT stream's work and that stream waits for them -- call it under `torch.cuda.stream(side)` to keep the
        compute stream free (the caller then records an event on `side` and makes the compute stream wait for it)."""
        import numpy as np

        arr = np.ascontiguousarray(arr)
        src = torch.from_numpy(arr)
        nbytes = arr.nbytes
        assert out.is_contiguous() and out.numel() * out.element_size() == nbytes
        if nbytes < (8 << 20) or os.environ.get("SCAMD_PINNED_UPLOAD") == "0" or src.is_pinned():
            out.copy_(src.view(out.dtype) if src.dtype != out.dtype else src, non_blocking=False)
            return
        with self._lock:
            stages, events, stream = self._state(out.device)
            out_b = out.view(torch.uint8).reshape(-1)
            src_b = arr.view(np.uint8).reshape(-1)
            cur = torch.cuda.current_stream(out.device)
            stream.wait_stream(cur)
            n_piece = (nbytes + self.PIECE - 1) // self.PIECE
            for i in range(n_piece):
                lo, hi = i * self.PIECE, min(nbytes, (i + 1) * self.PIECE)
                slot = i & 1
                stage = stages[slot]
                events[slot].synchronize()
                self._host_copy(stage.numpy()[:hi - lo], src_b[lo:hi])
                with torch.cuda.stream(stream):
                    out_b[lo:hi].copy_(stage[:hi - lo], non_blocking=True)
                    events[slot].record(stream)
            cur.wait_stream(stream)
            out.record_stream(stream)

    def download(self, t: torch.Tensor):
        """device tensor -> numpy array in pageable memory: DMA of piece i + 1 into one staging buffer while the thread
        pool copies piece i out of the other"""
        import numpy as np

        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        with self._lock:
            stages, events, stream = self._state(t.device)
            out = np.empty(tuple(t.shape), dtype=torch.empty((), dtype=t.dtype).numpy().dtype)
            out_b = out.view(np.uint8).reshape(-1)
            src_b = t.view(torch.uint8).reshape(-1)
            cur = torch.cuda.current_stream(t.device)
            stream.wait_stream(cur)  # `t` was produced on the current stream
            n_piece = (nbytes + self.PIECE - 1) // self.PIECE

            def issue(i):
                lo, hi = i * self.PIECE, min(nbytes, (i + 1) * self.PIECE)
                events[i & 1].synchronize()  # (an earlier upload may still be reading this staging buffer)
                with torch.cuda.stream(stream):
                    stages[i & 1][:hi - lo].copy_(src_b[lo:hi], non_blocking=True)
                    events[i & 1].record(stream)

            if n_piece:
                issue(0)
            for i in range(n_piece):
                lo, hi = i * self.PIECE, min(nbytes, (i + 1) * self.PIECE)
                events[i & 1].synchronize()  # piece i has arrived
                if i + 1 < n_piece:
                    issue(i + 1)  # (its staging buffer was drained in step i - 1)
                self._host_copy(out_b[lo:hi], stages[i & 1].numpy()[:hi - lo])
            t.record_stream(stream)
            return out


pinned_uploader = _PinnedTransfer()


_PINNED_RESULT_MAX = int(os.environ.get("SCAMD_PINNED_RESULT_MAX_MB", "256")) << 20


def to_host(t: torch.Tensor):
    """Device tensor -> numpy array.  `t.cpu()` into pageable memory uses the runtime's bounce buffer at less than half the
    PCIe rate, so results of 8 MB and more take one of two routes:
    * up to SCAMD_PINNED_RESULT_MAX_MB (default 256: the 120 MB of kNN distances and the 2 x 90 MB of connectivities at
      1M cells) they land in a PAGE-LOCKED block of torch's caching host allocator by one DMA -- the numpy array keeps the
      block alive, which bounds what one result can pin;
    * larger ones (10M cells: gigabytes) go through the staging pipeline of `_PinnedTransfer.download` into ordinary
      pageable memory: nothing stays page-locked beyond the two staging buffers (the pipeline costs ~4 ms per
      320 MB against the direct route -- first-touch page faults of the fresh array -- measured in bench.py's host-to-host
      leg: 116.8 vs 112.5 ms).
    SCAMD_PINNED_DOWNLOAD=0: plain `.cpu()`."""
    nbytes = t.numel() * t.element_size()
    if not t.is_cuda or nbytes < (8 << 20) or os.environ.get("SCAMD_PINNED_DOWNLOAD") == "0":
        return t.cpu().numpy()
    if nbytes <= _PINNED_RESULT_MAX:
        t = t.contiguous()
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host.numpy()
    return pinned_uploader.download(t)
