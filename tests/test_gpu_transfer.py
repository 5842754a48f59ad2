"""Host <-> HBM transfers of the drop-in path (scanpy_amd/_device.py: page-locked staging pipeline, per-device state;
ADVICE round 3): round trips are bit-exact at sizes that do not divide into pieces, results live in pageable memory."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize(("shape", "dtype"), [((9_000_001,), np.float32), ((1_000_003, 15), np.float64), ((33, 1_100_000), np.int32)])
def test_staged_upload_and_download_round_trip(shape, dtype, monkeypatch):
    import torch

    from scanpy_amd import _device

    rng = np.random.default_rng(0)
    a = rng.integers(-2**31, 2**31 - 1, size=int(np.prod(shape)), dtype=np.int64).astype(np.int32)
    a = np.resize(a.view(np.uint8), int(np.prod(shape)) * np.dtype(dtype).itemsize).view(dtype).reshape(shape)
    if np.issubdtype(dtype, np.floating):
        a = np.nan_to_num(a, nan=1.0, posinf=2.0, neginf=-2.0)
    dev = _device.require_gpu()
    t = _device.pinned_uploader.upload(a, dev)
    assert t.shape == tuple(shape) and t.device.type == "cuda"
    assert torch.equal(t.cpu(), torch.from_numpy(a))
    # results of up to SCAMD_PINNED_RESULT_MAX_MB: one DMA into a page-locked block
    back = _device.to_host(t)
    assert isinstance(back, np.ndarray) and back.dtype == a.dtype and np.array_equal(back.view(np.uint8), a.view(np.uint8))
    # larger ones: the staging pipeline, pageable memory
    monkeypatch.setattr(_device, "_PINNED_RESULT_MAX", 16 << 20)
    back2 = _device.to_host(t * 1)
    assert np.array_equal(back2.view(np.uint8), a.view(np.uint8))
    assert not torch.from_numpy(back2).is_pinned(), "large results must not keep host memory page-locked"
    assert len(_device.pinned_uploader._per_device) == 1  # the staging buffers of the device are re-used
