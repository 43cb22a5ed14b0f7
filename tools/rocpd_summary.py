"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max.
Usage: python tools/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def summarise(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
    q = f"""select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{name_col} order by 3 desc"""
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"| `{n[:90]}` | {c} | {t/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*t/total:.1f} |")
    return "\n".join(lines), cols


if __name__ == "__main__":
    text, _ = summarise(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)
