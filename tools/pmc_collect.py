"""Collect the per-kernel averages of the rocprofv3 --pmc passes of tools/profile_round2.sh into two small JSON files:
pmc_traffic.json {kernel: {FETCH_SIZE, WRITE_SIZE}} (KB per launch, as reported) and pmc_mfma.json {kernel: {counters...,
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES}}.   python tools/pmc_collect.py <dir with pmc_*/>"""
import glob
import json
import os
import sqlite3
import sys


def stamp():
    """What the counters were taken on: the source digest of the engine that is loaded (dtqn_build_info) and the commit the
    caller names in GIT_HEAD (the GPU box has no .git).  bench.py compares `src` with the engine it is timing."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dtqn_amd import engine
    info = engine.get_lib().dtqn_build_info().decode()
    return {"src": info.split("src=")[-1].split()[0], "git": os.environ.get("GIT_HEAD"), "tool": "tools/profile_round3.sh"}


def per_kernel(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    kc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in kc else "display_name"
    q = f"""select s.{name_col}, p.name, avg(e.value), count(*) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
            join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{name_col}, p.name"""
    out = {}
    for k, n, a, c in cur.execute(q):
        short = k.split("(")[0].replace("void dtqn::", "").replace("void ", "")
        if "dtqn" not in k and "tl_" not in k:
            continue
        out.setdefault(short, {})[n] = a
        out[short]["launches"] = c
    return out


def main(out):
    traffic, mfma = {}, {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for db in glob.glob(f"{out}/pmc_{c}/**/*results.db", recursive=True):
            for k, v in per_kernel(db).items():
                traffic.setdefault(k, {}).update({c: v.get(c)})
    for db in glob.glob(f"{out}/pmc_SQ_INSTS_VALU_MFMA_MOPS_F32/**/*results.db", recursive=True):
        for k, v in per_kernel(db).items():
            busy, tot = v.get("SQ_VALU_MFMA_BUSY_CYCLES"), v.get("SQ_BUSY_CYCLES")
            v["mfma_busy_frac"] = (busy / tot) if busy is not None and tot else None
            mfma[k] = v
    traffic["_meta"] = stamp()
    json.dump(traffic, open(f"{out}/pmc_traffic.json", "w"), indent=1, sort_keys=True)
    json.dump(mfma, open(f"{out}/pmc_mfma.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
