"""Run by tests/test_hdf5_write_cpu.py under an interpreter that HAS h5py (the image's /opt/conda/bin/python3.9): opens a
file scanpy_amd wrote with the HDF5 library itself and prints what it finds as JSON (TEST INFRASTRUCTURE)."""
import json
import sys

import h5py
import numpy as np


def describe(obj):
    out = {"attrs": {}}
    for k, v in obj.attrs.items():
        if isinstance(v, h5py.Empty):
            v = None
        elif isinstance(v, np.ndarray):
            v = [x.decode() if isinstance(x, bytes) else (x.item() if hasattr(x, "item") else x) for x in v.reshape(-1)]
        elif isinstance(v, bytes):
            v = v.decode()
        elif hasattr(v, "item"):
            v = v.item()
        out["attrs"][k] = v
    if isinstance(obj, h5py.Dataset):
        out.update(shape=list(obj.shape), dtype=str(obj.dtype), chunks=obj.chunks, compression=obj.compression,
                   shuffle=bool(obj.shuffle))
        data = obj[()]
        if obj.dtype.kind == "O":
            data = obj.asstr()[()]
            out["strings"] = data if isinstance(data, str) else [str(s) for s in np.asarray(data).reshape(-1)[:2000]]
        elif obj.dtype.names:
            out["fields"] = {n: np.asarray(data[n]).tolist() for n in obj.dtype.names}
        else:
            a = np.asarray(data)
            out["sum"] = float(a.astype(np.float64).sum()) if a.size else 0.0
            out["head"] = a.reshape(-1)[:8].tolist()
    else:
        out["children"] = {k: describe(obj[k]) for k in obj.keys()}
    return out


with h5py.File(sys.argv[1], "r") as f:
    json.dump(describe(f), sys.stdout)
