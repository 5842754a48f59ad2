"""Every `file.py:line[-line]` citation of the reference in the boundary documents points at an existing file and an
existing line range (guards against stale citations).  Needs /root/reference: skipped where it is absent (GPU box)."""
from __future__ import annotations

import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
DOCS = ["include/scanpy_amd.h", "INTEGRATION.md", "DESIGN.md"]
CITE = re.compile(r"(?<![\w/.])((?:src/scanpy|tests)/[\w/]+\.py):(\d+)(?:-(\d+))?")


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present")
@pytest.mark.parametrize("doc", DOCS)
def test_full_path_citations_resolve(doc):
    text = (ROOT / doc).read_text()
    cites = CITE.findall(text)
    assert cites, f"{doc}: no reference citations found"
    lengths = {}
    bad = []
    for path, lo, hi in cites:
        f = REF / path
        if not f.exists():
            bad.append(f"{path}: no such file")
            continue
        n = lengths.setdefault(path, len(f.read_text().splitlines()))
        lo_i, hi_i = int(lo), int(hi or lo)
        if not (1 <= lo_i <= hi_i <= n):
            bad.append(f"{path}:{lo}-{hi or lo} outside 1..{n}")
    assert not bad, "\n".join(bad)
