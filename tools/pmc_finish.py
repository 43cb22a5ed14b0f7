"""Finish a profile_round2.sh config directory: join the per-kernel MFMA counters (pmc_mfma.json) with the launch durations of
the kernel trace (kernel_stats.md) and derive the matrix-pipe busy fraction.   python tools/pmc_finish.py <cfg dir> <out json>"""
import json
import re
import sys

NOTE = ("rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES, averages per launch. "
        "On this stack the SQ counters come back per shader engine (8 CUs x 4 SIMDs): the MFMA count they imply is 1/32 of the "
        "instructions the kernels issue chip-wide.  mfma_busy_cycles_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / 32; mfma_util_vs_launch = "
        "that / (launch_us x 2400 cycles/us), i.e. the fraction of the launch during which an average SIMD's matrix pipe was busy, "
        "priced at the 2.4 GHz peak clock (a lower bound: the chip clocks lower under load).")


def main(cfg_dir, out):
    us = {}
    for line in open(f"{cfg_dir}/kernel_stats.md"):
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            us[m.group(1)] = float(m.group(4))
    d = json.load(open(f"{cfg_dir}/pmc_mfma.json"))
    res = {"_note": NOTE}
    for k, v in d.items():
        v = dict(v)
        v.pop("mfma_busy_frac", None)
        lu = us.get(k)
        if lu is None:       # the kernel-trace table truncates long (templated) names: match on the common prefix
            cands = [v_ for n_, v_ in us.items() if k.startswith(n_) or n_.startswith(k)]
            lu = cands[0] if len(cands) == 1 else None
        per = (v.get("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0) / 32.0
        v["launch_us"] = lu
        v["mfma_busy_cycles_per_simd"] = per
        v["mfma_busy_over_sq_busy_per_simd"] = per / v["SQ_BUSY_CYCLES"] if v.get("SQ_BUSY_CYCLES") else None
        v["mfma_util_vs_launch"] = per / (lu * 2400.0) if lu else None
        res[k] = v
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
