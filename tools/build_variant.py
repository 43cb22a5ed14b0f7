#!/usr/bin/env python3
"""Build an EXPERIMENT variant of the engine next to the product library:

    python tools/build_variant.py <tag> [-DFLAG=1 ...]     ->  tools/variants/libdtqn_hip_<tag>.so

Same sources and flags as dtqn_amd/build.py plus the given defines.  Load it with DTQN_HIP_LIB=<path> (dtqn_amd.engine);
the product path never looks here.  Used for A/B runs on the GPU box (the .so travels with the snapshot)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dtqn_amd import build as B      # noqa: E402


def main():
    tag, extra = sys.argv[1], sys.argv[2:]
    out = os.path.join(ROOT, "tools", "variants", f"libdtqn_hip_{tag}.so")
    objdir = os.path.join(ROOT, "tools", "variants", f"_obj_{tag}")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in B._sources():
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        cmd = ["hipcc", f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + os.path.join(ROOT, "include"),
               "-I" + B.CSRC, f'-DDTQN_BUILD_INFO="dtqn_hip {B.ARCH} variant={tag} {" ".join(extra)}"', s, "-o", o] + extra
        if s.endswith(".cpp"):
            cmd[1:2] = ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        outp, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"hipcc failed for {s}:\n{outp.decode()}")
    subprocess.check_call(["hipcc", f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs)
    subprocess.call(["rm", "-rf", objdir])
    print(out)


if __name__ == "__main__":
    main()
