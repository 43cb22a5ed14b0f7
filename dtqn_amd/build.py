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


def parse_resource_remarks(text: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {vgprs, agprs, scratch, lds, occupancy}}."""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:") or txt.startswith("Name:"):
            cur = out.setdefault(txt.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in txt:
            k, v = [x.strip() for x in txt.split(":", 1)]
            key = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "LDS Size [bytes/block]": "lds",
                   "Occupancy [waves/SIMD]": "occupancy", "SGPRs": "sgprs"}.get(k)
            if key is not None and v.lstrip("-").isdigit():
                cur[key] = int(v)
    return out


def resources_path() -> str:
    return LIB + ".resources.json"


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every source under dtqn_amd/csrc for gfx950 and link libdtqn_hip.so next to them."""
    prof = os.environ.get("DTQN_BUILD_PROF", "0") == "1"     # debug build with the stage clocks (tests/perf/stage_profile.py)
    tag = _digest() + ("+prof" if prof else "")
    stamp = LIB + ".stamp"
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == tag and os.path.exists(resources_path()):
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
        # (the resource remarks of every kernel are kept next to the library: tests/test_kernel_resources.py holds them against a budget)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + os.path.join(REPO, "include"),
               "-I" + CSRC, info, s, "-o", o] + (["-DDTQN_ENABLE_PROF"] if prof else []) + ["-Rpass-analysis=kernel-resource-usage"]
        if s.endswith(".cpp"):
            cmd[1:2] = []          # host-only C++ (may call the HIP runtime API): no offload arch needed
            cmd.insert(1, "-x")
            cmd.insert(2, "c++")
            cmd.insert(3, "-D__HIP_PLATFORM_AMD__")
            cmd.insert(4, "-I/opt/rocm/include")
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    resources = {}
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{out.decode()}")
        text = out.decode()
        for name, r in parse_resource_remarks(text).items():
            r["source"] = os.path.basename(s)
            resources[name] = r
        if verbose and out:
            print("\n".join(line for line in text.splitlines() if "-Rpass-analysis" not in line))
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    shutil.rmtree(objdir, ignore_errors=True)
    import json
    with open(resources_path(), "w") as f:
        json.dump({"src": tag, "kernels": resources}, f, indent=0, sort_keys=True)
    with open(stamp, "w") as f:
        f.write(tag)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
