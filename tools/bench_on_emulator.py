"""Dry run of bench.py with NO GPU: the kernels come from the host-emulated library (tests/emu/README.md), so what is
exercised is the bench's own control flow -- the JSON line, `parity`, `full_size_properties`, host-to-host -- on a matrix a
lane-by-lane executor can finish.  The numbers it prints mean nothing.  Test infrastructure, like everything under tests/emu.

    python tools/bench_on_emulator.py [bench.py arguments; default: 4000 x 300, weak structure]
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "emu"))
import patch_torch  # noqa: E402

patch_torch.activate()
import bench  # noqa: E402

if __name__ == "__main__":
    args = sys.argv[1:] or ["--n-obs", "4000", "--n-vars", "300", "--n-comps", "20", "--steps", "1", "--warmup", "0", "--cpu-sizes",
                            "2000,4000", "--h2h-reps", "1", "--no-side", "--no-noise-variant", "--structure", "weak"]
    sys.argv = ["bench.py", *args]
    bench.main()
