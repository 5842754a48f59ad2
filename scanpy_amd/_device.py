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
