"""Static scan of the gfx950 disassembly for global loads that are waited for where they are issued.

For every .hip file under scanpy_amd/csrc the device code is compiled to assembly (no GPU needed) and every kernel is
searched for `global_load*` instructions INSIDE A LOOP that are followed, within three instructions, by
`s_waitcnt vmcnt(0)`: a dependent-gather chain with one request in flight per wave.  Prints the kernels with the most
such sites.  Not every hit matters (a short loop may be covered by other resident waves -- the kNN select kernel's
per-cell prologue is one), but every latency-bound kernel found in round 2 was on this list.

    python tools/isa_wait_scan.py [file.hip ...]
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-pass-failed", "-x", "hip", "-S",
         "--cuda-device-only", f"-I{ROOT / 'include'}", f"-I{ROOT / 'scanpy_amd' / 'csrc'}"]


def scan(asm: str) -> dict[str, int]:
    hits: dict[str, int] = {}
    cur, in_loop = None, False
    lines = asm.split("\n")
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, in_loop = m.group(1), False
        if "Loop Header" in line:
            in_loop = True
        if "s_endpgm" in line:
            in_loop = False
        if cur and in_loop and "global_load" in line:
            seen = 0
            for nxt in lines[i + 1:i + 10]:
                t = nxt.strip()
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                seen += 1
                if "s_waitcnt" in t and "vmcnt(0)" in t:
                    hits[cur] = hits.get(cur, 0) + 1
                    break
                if "global_load" in t or seen >= 3:
                    break
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
        demangle = subprocess.run(["c++filt", *hits.keys()], capture_output=True, text=True).stdout.split("\n") if hits else []
        names = dict(zip(hits.keys(), demangle))
        for k, v in sorted(hits.items(), key=lambda kv: -kv[1])[:12]:
            print(f"{f.name:16s} {v:3d}  {names.get(k, k)[:110]}")


if __name__ == "__main__":
    main()
