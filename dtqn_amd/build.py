"""Build libdtqn_hip.so (the gfx950 engine) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(CSRC, "libdtqn_hip.so")
ARCH = "gfx950"


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest() -> str:
    h = hashlib.sha1()
    deps = _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp")))
    deps.append(os.path.join(REPO, "include", "dtqn_hip.h"))
    for d in deps:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every source under dtqn_amd/csrc for gfx950 and link libdtqn_hip.so next to them."""
    prof = os.environ.get("DTQN_BUILD_PROF", "0") == "1"     # debug build with the stage clocks (tests/perf/stage_profile.py)
    tag = _digest() + ("+prof" if prof else "")
    stamp = LIB + ".stamp"
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == tag:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 engine")
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    info = f'-DDTQN_BUILD_INFO="dtqn_hip {ARCH} src={tag}"'
    procs, objs = [], []
    for s in _sources():
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + os.path.join(REPO, "include"),
               "-I" + CSRC, info, s, "-o", o] + (["-DDTQN_ENABLE_PROF"] if prof else [])
        if s.endswith(".cpp"):
            cmd[1:2] = []          # host-only C++ (may call the HIP runtime API): no offload arch needed
            cmd.insert(1, "-x")
            cmd.insert(2, "c++")
            cmd.insert(3, "-D__HIP_PLATFORM_AMD__")
            cmd.insert(4, "-I/opt/rocm/include")
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    shutil.rmtree(objdir, ignore_errors=True)
    with open(stamp, "w") as f:
        f.write(tag)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
