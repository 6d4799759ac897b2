"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into a per-kernel stats CSV (name, calls, total/avg/min/max us, %).
    python tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
rows = list(db.execute(q))
tot = sum(r[2] for r in rows)
lines = ["Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage"]
for r in rows:
    lines.append('"%s",%d,%.2f,%.3f,%.3f,%.3f,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
else:
    sys.stdout.write(out)
