"""Per-dispatch durations of the kernels whose name contains a pattern, in launch order (rocprofv3 rocpd sqlite trace).
Usage: python tools/rocpd_sequence.py <results.db> <pattern> [max_rows]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
rows = list(cur.execute(f"""select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                            where s.{name_col} like ? order by d.start""", (f"%{sys.argv[2]}%",)))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
print(" ".join(f"{(e - s) / 1e3:.1f}" for _, s, e in rows[-n:]))
