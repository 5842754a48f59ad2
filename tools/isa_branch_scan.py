"""Static scan of the gfx950 disassembly for INNERMOST loops whose body contains unconditional branches.

A taken branch costs a wave ~16 cycles of instruction fetch -- as much as four scalar instructions.  hipcc lays the body of
an `if` it believes unlikely OUT of line: the hot path then jumps there and back (`s_branch`) on every trip.  In the kNN select
kernel's list insertion that was one taken branch per survivor in a 26-instruction chain; `__builtin_expect(cond, 1)` on the
guard put the body in line: 11.8 -> 11.45 ms (profiles/r05w2_*).  This prints, per kernel, the innermost loops with the most
`s_branch` instructions among their blocks and the loop's length -- candidates, not findings: whether a site matters is a
question for an A/B on the GPU.

    python tools/isa_branch_scan.py [file.hip ...]
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


def scan(asm: str):
    """-> list of (kernel, loop header label, depth, instructions in the loop's blocks, s_branch among them, cond branches)"""
    out = []
    cur = None
    block_loop = None  # header label of the innermost loop the current block belongs to
    loops: dict[tuple[str, str], dict] = {}
    inner_headers: set[tuple[str, str]] = set()
    lines = asm.split("\n")
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, block_loop = m.group(1), None
            continue
        if cur is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:|^; %bb\.\d+:", line):
            lab = re.match(r"^(\.LBB\d+_\d+):", line)
            # the comment lines that follow a block label say which loop it is in
            text = " ".join(lines[i:i + 8])
            hdr = None
            if "This Inner Loop Header" in text.split("s_", 1)[0] and lab:
                hdr = lab.group(1)[1:]
                d = re.search(r"This Inner Loop Header: Depth=(\d+)", text)
                loops[(cur, hdr)] = {"depth": int(d.group(1)) if d else 0, "n": 0, "br": 0, "cbr": 0}
                inner_headers.add((cur, hdr))
            else:
                mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", line + " " + (lines[i + 1] if i + 1 < len(lines) else ""))
                if mm:
                    hdr = mm.group(1)
            block_loop = (cur, "LBB" + hdr[2:] if hdr and hdr.startswith("BB") else hdr) if hdr else None
            continue
        if "s_endpgm" in line:
            block_loop = None
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if block_loop in loops:
            rec = loops[block_loop]
            rec["n"] += 1
            if t.startswith("s_branch"):
                rec["br"] += 1
            elif t.startswith("s_cbranch"):
                rec["cbr"] += 1
    for (k, h), rec in loops.items():
        if rec["br"] > 0:
            out.append((k, h, rec["depth"], rec["n"], rec["br"], rec["cbr"]))
    return out


def main():
    files = [Path(a) for a in sys.argv[1:]] or sorted((ROOT / "scanpy_amd" / "csrc").glob("*.hip"))
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            s = Path(td) / (f.stem + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(f), "-o", str(s)], check=True, stderr=subprocess.DEVNULL)
            hits = scan(s.read_text())
            if not hits:
                continue
            print(f"== {f.name}")
            per_kernel: dict[str, list] = {}
            for k, h, d, n, br, cbr in hits:
                per_kernel.setdefault(k, []).append((h, d, n, br, cbr))
            for k, lst in sorted(per_kernel.items(), key=lambda kv: -sum(x[3] for x in kv[1])):
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
                tot = sum(x[3] for x in lst)
                short = sorted(lst, key=lambda x: x[2])[:4]
                print(f"  {tot:4d} s_branch in {len(lst):3d} innermost loops  {name}")
                for h, d, n, br, cbr in short:
                    print(f"         {h}: depth {d}, {n} instructions, {br} s_branch, {cbr} s_cbranch")


if __name__ == "__main__":
    main()
