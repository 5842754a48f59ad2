"""The few settings the path reads (src/scanpy/_settings/__init__.py:59-218): N_PCS, n_jobs, verbosity."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Settings:
    N_PCS: int = 50        # _settings/__init__.py:83
    n_jobs: int = 4        # _settings/__init__.py:132 (unused on the GPU path; kept for API parity)
    verbosity: int = 1
    # Default Leiden flavor.  The reference's V1 preset says 'leidenalg' (presets.py:271-277); both flavors
    # optimise the same objective and map onto the same GPU kernel here.
    leiden_flavor: str = "leidenalg"
    # cells every query block probes when `pp.neighbors(transformer='ivf')` asks for the approximate search
    # (scamd_knn_l2_ivf_f32; ~2048 rows per cell).  The recall / work curve is data dependent: DESIGN.md 3.1.
    knn_nprobe: int = 32
    # `sc.settings.preset` (src/scanpy/_settings/presets.py:179-188), as far as the path reads it: 'ScanpyV1' (default) or
    # 'ScanpyV2Preview'.  It decides how the igraph flavor of `tl.leiden` builds its graph (src/scanpy/_utils/__init__.py:
    # 285-298): V1 adds every stored entry of the symmetric matrix as an undirected edge (every pair twice), V2 every
    # pair once -- which matters for the CPM objective only (tools/_leiden.py).
    preset: str = "ScanpyV1"


settings = Settings()
