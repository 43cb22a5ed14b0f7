#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in the given HIP sources (hipcc -Rpass-analysis=kernel-resource-usage,
gfx950, the product build's flags).   python tools/kernel_resources.py dtqn_amd/csrc/dtqn_forward.hip [more.hip] [-DFLAG]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(src, extra):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "dtqn_amd", "csrc"), "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if "error:" in line:
            print(line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:") or txt.startswith("Name:"):
            cur = {"name": txt.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def short(name):
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(.*\)$", "", d)
    return d.replace("dtqn::", "")


if __name__ == "__main__":
    srcs = [a for a in sys.argv[1:] if not a.startswith("-")]
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    print("| kernel | VGPRs | AGPRs | scratch B/lane | LDS B | occupancy waves/SIMD |\n|---|---|---|---|---|---|")
    for s in srcs:
        for r in table(s, extra):
            print(f"| {short(r['name'])} | {r.get('VGPRs', '?')} | {r.get('AGPRs', '?')} | {r.get('ScratchSize [bytes/lane]', '?')} | "
                  f"{r.get('LDS Size [bytes/block]', '?')} | {r.get('Occupancy [waves/SIMD]', '?')} |")
