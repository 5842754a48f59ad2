"""TEST INFRASTRUCTURE: builds `tests/emu/_build/libscanpy_amd_emu.so` -- the kernel sources of scanpy_amd/csrc compiled
for the HOST against tests/emu/hip/hip_runtime.h (lane-by-lane executor, tests/emu/README.md).  The product never loads
this library; `scanpy_amd/_lib.py` knows only `scanpy_amd/_lib/libscanpy_amd.so`.

The sources are used as they are, except for four constructs that have no host spelling and are rewritten on a copy:
  * `extern __shared__ T name[];`                      -> a pointer to the executor's dynamic-LDS block
  * the `llvm.amdgcn.writelane` declaration (knn.hip) -> its one-line meaning
  * two `asm volatile("s_getreg_b32 ...")` probes     -> 0
  * `Workspace::take` (common.h)                       -> the same carving plus a poisoned gap after every buffer when
                                                          the build is an AddressSanitizer one (EMU_ASAN=1)
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "scanpy_amd" / "csrc"
CLANG = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")

DYN_LDS = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w ]*?)\s+(\w+)\[\];")


def transform(name: str, text: str, asan: bool) -> str:
    text = DYN_LDS.sub(lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(::emu::dyn_lds());", text)
    if name == "knn.hip":
        text, n = re.subn(r'extern "C" __device__ int scamd_llvm_writelane\(int value, int lane, int old\) __asm\("llvm\.amdgcn\.writelane\.i32"\);',
                          "static inline int scamd_llvm_writelane(int value, int lane, int old) { return ::emu::g_cur->lane == lane ? value : old; }", text)
        assert n == 1, "writelane declaration not found"
        text, n = re.subn(r'asm volatile\("s_getreg_b32 %0, hwreg\((\w+)\)" : "=s"\((\w+)\)\);', r"\2 = 0;", text)
        assert n == 2, "s_getreg probes not found"
    if name == "common.h" and asan:
        old = "    off += bytes;\n    if (base && off > cap) ok = false;\n    return p;"
        assert old in text
        text = text.replace(old, "    off += bytes;\n    if (base && off > cap) ok = false;\n"
                                 "    if (base && ok) ::emu_poison_gap(base + off, align_up(off, 256) - off);\n    return p;")
        text = text.replace("namespace scamd {", 'extern "C" void emu_poison_gap(void* p, size_t n);\nextern "C" void emu_unpoison(void* p, size_t n);\nnamespace scamd {', 1)
        ctor = "Workspace(void* b, size_t c) : base(static_cast<char*>(b)), cap(c) {}"
        assert ctor in text
        text = text.replace(ctor, "Workspace(void* b, size_t c) : base(static_cast<char*>(b)), cap(c) { if (base) ::emu_unpoison(base, cap); }")
    return text


def build(asan: bool = False, force: bool = False) -> Path:
    tag = "asan" if asan else "plain"
    out_dir = HERE / "_build" / tag
    lib = out_dir / "libscanpy_amd_emu.so"
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cpp"))
    deps = srcs + sorted(CSRC.glob("*.h")) + [HERE / "hip" / "hip_runtime.h", HERE / "emu_runtime.cpp", Path(__file__),
                                               ROOT / "include" / "scanpy_amd.h"]
    if lib.exists() and not force and lib.stat().st_mtime > max(d.stat().st_mtime for d in deps):
        return lib
    work = out_dir / "x" / "csrc"
    work.mkdir(parents=True, exist_ok=True)
    (out_dir / "include").mkdir(exist_ok=True)
    shutil.copy(ROOT / "include" / "scanpy_amd.h", out_dir / "include" / "scanpy_amd.h")
    for f in list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")) + list(CSRC.glob("*.h")):
        (work / f.name).write_text(transform(f.name, f.read_text(), asan))
    # -Og, not -O1: at -O1 clang clones a rendezvous into the two arms of a lane-dependent branch although the callee is
    # declared convergent + noduplicate (seen in chol_factor_kernel: 18 instead of 12 calls) -- two call sites, and the
    # lanes of the two arms stop meeting
    flags = ["-std=c++17", "-Og", "-g", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-everything", f"-I{HERE}", "-DSCAMD_EMU=1"]
    if asan:
        flags += ["-fsanitize=address", "-fno-omit-frame-pointer", "-shared-libasan"]
    objs = []
    procs = []
    for f in srcs:
        o = out_dir / (f.stem + ".o")
        objs.append(o)
        procs.append((f.name, subprocess.Popen([CLANG, "-x", "c++", *flags, "-c", str(work / f.name), "-o", str(o)],
                                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    o = out_dir / "emu_runtime.o"
    objs.append(o)
    procs.append(("emu_runtime.cpp", subprocess.Popen([CLANG, *flags, "-c", str(HERE / "emu_runtime.cpp"), "-o", str(o)],
                                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"---- {name}\n{out[-6000:]}\n")
    if failed:
        raise RuntimeError("emulator build failed")
    subprocess.run([CLANG, "-shared", *(["-fsanitize=address", "-shared-libasan"] if asan else []), *map(str, objs), "-o", str(lib)], check=True)
    return lib


if __name__ == "__main__":
    print(build(asan="--asan" in sys.argv, force="--force" in sys.argv))
