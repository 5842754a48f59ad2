"""TEST INFRASTRUCTURE: run the `-m gpu` tests against the HOST-EMULATED kernel library (tests/emu/README.md).

`SCAMD_TESTS_ON_EMULATOR=1 python -m pytest tests -m gpu -k ...` -- tests/conftest.py then calls `activate()` before
anything imports scanpy_amd: the C ABI of the emulated library takes host pointers, so "device" tensors are CPU tensors
that claim to be CUDA ones.  This is a way to exercise the kernels' logic (and, with SCAMD_EMU_ASAN=1 under an
LD_PRELOADed ASan runtime, their indexing) when no GPU minute is left; it is never active in a normal run, the product
has no such switch, and a test that passes here has said nothing about the GPU build."""
from __future__ import annotations

import os
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent


def activate() -> None:
    import torch

    sys.path.insert(0, str(HERE))
    import harness

    emu = harness.load(asan=os.environ.get("SCAMD_EMU_ASAN") == "1")

    class _Stream:
        cuda_stream = 0

        def synchronize(self):
            pass

        def wait_stream(self, other):
            pass

        def wait_event(self, event):
            pass

    cpu = torch.device("cpu")
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.current_device = lambda: 0
    torch.cuda.device_count = lambda: 1
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.mem_get_info = lambda *a, **k: (64 << 30, 64 << 30)
    torch.Tensor.cuda = lambda self, *a, **k: self

    class _Event:
        def __init__(self, *a, **k):
            pass

        def record(self, *a, **k):
            pass

        def synchronize(self):
            pass

        def wait(self, *a, **k):
            pass

        def elapsed_time(self, other):
            return 0.0

    class _StreamCtx(_Stream):
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    torch.cuda.Stream = _StreamCtx
    torch.cuda.Event = _Event
    torch.cuda.stream = lambda s: _StreamCtx()
    _orig_device = torch.device

    class _DeviceMeta(type):
        def __instancecheck__(cls, obj):
            return isinstance(obj, _orig_device)

    class _Device(metaclass=_DeviceMeta):  # torch.device("cuda"[, i]) -> the cpu device; everything else unchanged
        def __new__(cls, *a, **k):
            if a and isinstance(a[0], str) and a[0].startswith("cuda"):
                return cpu
            return _orig_device(*a, **k)

    torch.device = _Device
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        a = tuple(cpu if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = cpu
        return _orig_to(self, *a, **k)

    torch.Tensor.to = _to
    _orig_empty = torch.empty

    def _empty(*a, pin_memory=False, **k):  # (no page-locked memory without a GPU)
        return _orig_empty(*a, **k)

    torch.empty = _empty
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.Tensor.record_stream = lambda self, s: None

    from scanpy_amd import _device, _lib

    import numpy as np

    import scanpy_amd  # noqa: F401  (imports every submodule)

    patched = {
        "require_gpu": lambda: cpu,
        "set_device_from_env": lambda: cpu,
        "stream_ptr": lambda: None,
        "to_host": lambda t: t.detach().numpy(),
    }
    _lib.load = lambda: emu
    # ASan build: the gaps between the carve-outs of a call stay poisoned in the pooled workspace; entry points that do
    # not carve (scamd_colsum_f32_f64 takes the raw pointer) would trip over the previous call's gaps
    from scanpy_amd import _kernels

    _orig_ws = _kernels._ws

    def _ws(nbytes, dev):
        import ctypes as C

        buf, size = _orig_ws(nbytes, dev)
        emu.emu_unpoison(C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()))
        return buf, size

    _kernels._ws = _ws
    _device.pinned_uploader.upload = lambda arr, device: torch.from_numpy(np.ascontiguousarray(arr))
    # (`from ._device import require_gpu, ...` copied the originals into the importing modules)
    for name, mod in list(sys.modules.items()):
        if name == "scanpy_amd" or name.startswith("scanpy_amd."):
            for attr, fn in patched.items():
                if hasattr(mod, attr):
                    setattr(mod, attr, fn)
