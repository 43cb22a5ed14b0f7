#!/bin/bash
# rocprofv3 passes behind profiles/r01_*: kernel trace + stats of bench.py, then FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (MI355X_MICROARCH.md: counters in their own runs, kernel-trace only).  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/{kernel_stats.md, bench_under_rocprof.json, pmc_traffic.json}
set -u
TAG=${1:-r01_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-env-rate --no-other-configs"
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $BENCH > $OUT/bench_kt.log 2>&1
grep '^{"metric"' $OUT/bench_kt.log | tail -1 > $OUT/bench_under_rocprof.json
DB=$(find $OUT/kt -name '*results.db' | head -1)
python tools/rocpd_summary.py "$DB" $OUT/kernel_stats.md > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -- $BENCH > $OUT/bench_pmc_$C.log 2>&1
done
python - "$OUT" <<'PY'
import sqlite3, sys, json, glob
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(f"{out}/pmc_{c}/**/*results.db", recursive=True)
    if not dbs:
        continue
    con = sqlite3.connect(dbs[0]); cur = con.cursor()
    kc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in kc else "display_name"
    q = f"""select s.{name_col}, p.name, avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
            join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            where s.{name_col} like '%dtqn%' group by s.{name_col}, p.name"""
    for k, n, a in cur.execute(q):
        short = k.split("(")[0].split("<")[0].replace("void dtqn::", "")
        res.setdefault(short, {})[n] = a
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat $OUT/kernel_stats.md | head -12
cat $OUT/bench_under_rocprof.json | cut -c1-400
