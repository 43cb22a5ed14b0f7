"""Kernel-resource gate: no product kernel may exceed its scratch budget (tests/kernel_budget.json).

dtqn_amd.build compiles every translation unit with -Rpass-analysis=kernel-resource-usage and keeps the per-kernel table next to the
library (libdtqn_hip.so.resources.json, stamped with the source digest).  Round 5 shipped a GRU-gated backward instantiation that had
silently grown from 664 to 4 572 bytes of scratch per lane; this test is the gate against a repeat.  Compile-only: no GPU needed."""
import json
import os
import re
import subprocess

import pytest

from dtqn_amd import build as B

HERE = os.path.dirname(os.path.abspath(__file__))


def _short(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.splitlines()
    res = []
    for d in out:
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*\)$", "", d)
        res.append(d.replace("dtqn::", ""))
    return res


@pytest.fixture(scope="module")
def resources():
    path = B.resources_path()
    stale = True
    if os.path.exists(path):
        with open(path) as f:
            stale = json.load(f).get("src", "").split("+")[0] != B._digest()
    if stale:
        B.build()                      # (a few minutes on 8 cores; a no-op when __graft_entry__.build() ran on this tree)
    with open(path) as f:
        data = json.load(f)
    assert data["src"].split("+")[0] == B._digest(), "resource table does not belong to this source tree"
    return data["kernels"]


def test_remark_parser():
    text = ("a.hip:5:1: remark: Function Name: _ZN4dtqn1kEv [-Rpass-analysis=kernel-resource-usage]\n"
            "a.hip:5:1: remark:     VGPRs: 12 [-Rpass-analysis=kernel-resource-usage]\n"
            "a.hip:5:1: remark:     ScratchSize [bytes/lane]: 48 [-Rpass-analysis=kernel-resource-usage]\n"
            "a.hip:5:1: remark:     Occupancy [waves/SIMD]: 8 [-Rpass-analysis=kernel-resource-usage]\n"
            "a.hip:5:1: remark:     LDS Size [bytes/block]: 1024 [-Rpass-analysis=kernel-resource-usage]\n")
    assert B.parse_resource_remarks(text) == {"_ZN4dtqn1kEv": {"vgprs": 12, "scratch": 48, "occupancy": 8, "lds": 1024}}


def test_every_kernel_inside_its_scratch_budget(resources):
    with open(os.path.join(HERE, "kernel_budget.json")) as f:
        budget = json.load(f)
    assert len(resources) > 250, "the table should hold every instantiation of the engine"
    mangled = sorted(resources)
    over, used = [], set()
    for m, name in zip(mangled, _short(mangled)):
        limit, why = budget["default_scratch_max"], "default"
        for i, rule in enumerate(budget["rules"]):
            if re.search(rule["match"], name):
                limit, why = rule["scratch_max"], rule["why"]
                used.add(i)
                break
        scratch = resources[m].get("scratch")
        assert scratch is not None, f"no scratch figure for {name}"
        if scratch > limit:
            over.append(f"{name}: {scratch} B/lane > {limit} ({why}) [{resources[m].get('source')}]")
    assert not over, "kernels over their scratch budget:\n" + "\n".join(over)
    unused = [budget["rules"][i]["match"] for i in range(len(budget["rules"])) if i not in used]
    assert not unused, f"budget rules that match no kernel (renamed instantiation?): {unused}"
