"""Static screen of the gfx950 disassembly for cross-lane reads that execute under a NARROWED exec mask.

`ds_bpermute_b32` / `ds_permute_b32` / `ds_swizzle_b32` / DPP moves read another lane's register; a source lane that is
switched off by EXEC contributes nothing, and what the reader then gets is not defined by HIP (round 2's Leiden was not
reproducible because of ONE such site: a `__shfl` inside `cond ? a : __shfl(...)`).  For every kernel of
every .hip file the assembly (compiled with line info) is walked linearly, counting exec-narrowing instructions
(`s_and_saveexec`, `s_andn2 exec`) against widening ones (`s_or exec`, `s_mov exec, -1`); cross-lane reads met at depth
> 0 are listed with their source line.  A hit is a site to READ, not a bug by itself: a kernel-wide `if (v < n)` guard
also narrows exec, and a shuffle among the surviving lanes is fine.

    python tools/isa_shuffle_scan.py [file.hip ...]
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["-O3", "-g1", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-pass-failed", "-x", "hip", "-S",
         "--cuda-device-only", f"-I{ROOT / 'include'}", f"-I{ROOT / 'scanpy_amd' / 'csrc'}"]
CROSS = re.compile(r"\b(ds_bpermute_b32|ds_permute_b32|ds_swizzle_b32)\b|\b(row_shr|row_shl|row_ror|wave_shr|wave_shl|row_bcast|quad_perm|row_mirror|row_half_mirror|row_newbcast)")


def scan(asm: str):
    hits = []
    cur, depth, loc = None, 0, ""
    files = {}
    for line in asm.split("\n"):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, depth = m.group(1), 0
            continue
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", t)
        if m:
            files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = f"{files.get(m.group(1), m.group(1))}:{m.group(2)}"
            continue
        if not cur or not t or t.startswith(";"):
            continue
        if "s_endpgm" in t:
            depth = 0
        elif re.match(r"s_(and|andn2)_saveexec_b64", t) or re.match(r"s_(and|andn2)_b64\s+exec", t):
            depth += 1
        elif re.match(r"s_or_b64\s+exec", t) or re.match(r"s_or_saveexec_b64", t):
            depth = max(0, depth - 1)
        elif re.match(r"s_mov_b64\s+exec,\s*-1", t):
            depth = 0
        elif depth > 0 and CROSS.search(t):
            hits.append((cur, loc, depth, t.split(";")[0].strip()))
    return hits


def main() -> None:
    files = [Path(a) for a in sys.argv[1:]] or sorted((ROOT / "scanpy_amd" / "csrc").glob("*.hip"))
    for f in files:
        with tempfile.TemporaryDirectory() as td:
            out = Path(td) / "k.s"
            r = subprocess.run(["hipcc", *FLAGS, str(f), "-o", str(out)], capture_output=True, text=True)
            if r.returncode != 0 or not out.exists():
                print(f"{f.name}: compile failed\n{r.stderr[-400:]}")
                continue
            hits = scan(out.read_text())
        by_site = {}
        for k, loc, depth, ins in hits:
            by_site.setdefault((k, loc), []).append(ins.split()[0])
        print(f"{f.name}: {len(by_site)} source sites with cross-lane reads under a narrowed exec mask")
        for (k, loc), ins in sorted(by_site.items()):
            kn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"   {loc:<22} {kn:<60} {len(ins)} x {sorted(set(ins))}")


if __name__ == "__main__":
    main()
