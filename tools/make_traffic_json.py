"""profiles/knn_select_traffic.json from the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the select kernel.
    python tools/make_traffic_json.py gpurun_out/r04c/knn_xcd_pmc_FETCH_SIZE.csv gpurun_out/r04c/knn_xcd_pmc_WRITE_SIZE.csv"""
from __future__ import annotations

import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def total(path: str, counter: str):
    rows = [r for r in csv.DictReader(open(path)) if "knn_select_reg" in r.get("Kernel_Name", "") and r["Counter_Name"] == counter]
    gmax = max(int(r["Grid_Size"]) for r in rows)
    rows = [r for r in rows if int(r["Grid_Size"]) == gmax]
    return sum(float(r["Counter_Value"]) for r in rows), rows[0]["Kernel_Name"], gmax


def main():
    fetch, name, grid = total(sys.argv[1], "FETCH_SIZE")
    write, _, _ = total(sys.argv[2], "WRITE_SIZE")
    out = {
        "kernel": name, "engine": "bf16x3" if ("true, true" in name or (", true>" in name and name.count("true") == 2)) else "f32",
        "mode": "ivf", "grid_threads": grid, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
        # raw = the counters as reported; corrected = FETCH_SIZE doubled (MI355X_MICROARCH.md, "HBM [CDNA4]": on gfx950
        # FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read -- 16 B per lane, global_load and
        # buffer_load ... lds alike, which is what this kernel's tile stream is; WRITE_SIZE is uncalibrated and left as is)
        "bytes_per_launch_raw": (fetch + write) * 1024.0,
        "bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "fetch_correction": {"factor": 2.0, "source": "/opt/skills/guides/MI355X_MICROARCH.md, section 'HBM [CDNA4]'"},
        "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counters restricted to the select kernel: --kernel-include-regex) "
                "of tools/knn_only.py 1000000 1 on the bench's own embedding (cell-pruned sweep, XCD-aware launch order, one launch, summed "
                "over the dispatch's rows = all XCDs); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced "
                "reads); L2-side fabric requests incl. Infinity-Cache hits",
    }
    (ROOT / "profiles" / "knn_select_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
