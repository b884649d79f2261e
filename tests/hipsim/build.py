"""Build tests/hipsim/_build/libtjm_sim.so: the .hip sources of yaqs_amd/csrc compiled for the HOST against tests/hipsim/hip/hip_runtime.h.

TEST INFRASTRUCTURE ONLY (see the header of hip/hip_runtime.h).  The sources are taken as they are; the one construct the host
compiler cannot take from a macro, ``extern __shared__ T name[];`` (the dynamic LDS segment), is rewritten on a scratch copy into
``#define name ((T*)hipsim::dyn_shared())``.

    python tests/hipsim/build.py            # incremental
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
SRC = os.path.join(ROOT, "yaqs_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libtjm_sim.so")
CXX = os.environ.get("HIPSIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g1", "-fPIC", "-ffp-contract=off", "-pthread", "-I", HERE, "-Wno-unused-value",
         "-Wno-unknown-attributes", "-Wno-unused-function"]
# store callbacks for the kernel sources (hipsim_rt.cpp: "Lock step"): the compiler's address-sanitizer instrumentation reduced to
# one call per store; the callbacks live in hipsim_rt.cpp, the sanitizer's runtime is not linked
HOOKS = ["-fsanitize=address", "-mllvm", "-asan-instrument-reads=0", "-mllvm", "-asan-instrumentation-with-call-threshold=0", "-mllvm", "-asan-stack=0",
         "-mllvm", "-asan-globals=0", "-mllvm", "-asan-use-after-return=never", "-mllvm", "-asan-detect-invalid-pointer-pair=0"]
DYN = re.compile(r"extern\s+__shared__\s+(\w+)\s+(\w+)\s*\[\s*\]\s*;")


def _write_if_changed(path: str, text: str) -> None:
    if os.path.exists(path):
        with open(path) as f:
            if f.read() == text:
                return
    with open(path, "w") as f:
        f.write(text)


def build(verbose: bool = False, f32: bool = False) -> str:
    """``f32``: the complex64 variant of the sources (-DTJM_F32, libtjm_sim_f32.so), objects in their own directory."""
    if f32:
        return _build(verbose, os.path.join(OUT, "f32"), os.path.join(OUT, "libtjm_sim_f32.so"), ["-DTJM_F32"])
    return _build(verbose, OUT, LIB, [])


def _build(verbose: bool, OUT: str, LIB: str, extra: list) -> str:
    csrc = os.path.join(OUT, "yaqs_amd", "csrc")
    os.makedirs(csrc, exist_ok=True)
    os.makedirs(os.path.join(OUT, "include"), exist_ok=True)
    shim = os.path.join(HERE, "hip", "hip_runtime.h")
    with open(os.path.join(ROOT, "include", "tjm_hip.h")) as f:
        _write_if_changed(os.path.join(OUT, "include", "tjm_hip.h"), f.read())
    for name in sorted(os.listdir(SRC)):
        if name.endswith(".hip") or name.endswith(".h"):
            with open(os.path.join(SRC, name)) as f:
                text = DYN.sub(lambda m: f"#define {m.group(2)} (({m.group(1)}*)hipsim::dyn_shared())", f.read())
            _write_if_changed(os.path.join(csrc, name), text)
    headers = [shim, os.path.join(OUT, "include", "tjm_hip.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    stamp = max(os.path.getmtime(h) for h in headers)
    units = [(os.path.join(csrc, f), os.path.join(csrc, f[:-4] + ".o")) for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
    units.append((os.path.join(HERE, "hipsim_rt.cpp"), os.path.join(OUT, "hipsim_rt.o")))
    jobs = [(s, o) for s, o in units if not os.path.exists(o) or os.path.getmtime(o) < max(stamp, os.path.getmtime(s))]
    procs = []
    for src, obj in jobs:
        cmd = [CXX] + FLAGS + extra + (HOOKS if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src}\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipsim build failed")
    if jobs or not os.path.exists(LIB):
        subprocess.check_call([CXX, "-shared", "-fPIC", "-pthread"] + [o for _, o in units] + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
