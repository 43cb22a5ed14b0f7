"""Average PMC counters per kernel from a rocprofv3 rocpd database.  usage: pmc_summary.py <db> [kernel-substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else "dtqn"
kc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kc else "display_name"
pc = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
ic = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
q = f"""select s.{name_col}, p.name, count(*), avg(e.value), sum(e.value)
        from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
        join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        where s.{name_col} like '%{sub}%' group by s.{name_col}, p.name order by 1, 2"""
try:
    rows = list(cur.execute(q))
except Exception as ex:
    print("query failed:", ex); print("pmc_event cols", pc); print("info_pmc cols", ic); raise
last = None
for k, n, c, a, t in rows:
    if k != last:
        print(f"\n{k[:100]}  (dispatches: {c})"); last = k
    print(f"  {n:32s} avg/dispatch {a:16.1f}")
