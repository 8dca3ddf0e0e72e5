"""rocprofv3's rocpd database (the default output of `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- ...`)
-> the per-kernel summary kept under profiles/ (name, calls, total / average / min / max duration in ns, share)."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = cur.execute('select name, count(*), sum("end"-start), avg("end"-start), min("end"-start), max("end"-start) '
                   "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name, calls, tot, f"{avg:.1f}", f"{100 * tot / total:.2f}", mn, mx])
for r in rows[:12]:
    print(f"{r[0][:72]:72s} {r[1]:6d} calls  avg {r[3] / 1e3:10.2f} us  {100 * r[2] / total:5.1f} %")
