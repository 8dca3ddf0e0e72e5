"""rocprofv3 --pmc database -> one row per (kernel, counter): dispatches, mean value.  Run ON the GPU box right
after the profile (the databases of whole-bench runs are too large to bring back)."""
import csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
only = sys.argv[3:] or None
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                   "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "mean_value", "mean_duration_ns"])
    for k, c, n, v, d in rows:
        if only and not any(o in k for o in only):
            continue
        w.writerow([k[:90], c, n, f"{v:.1f}", f"{d:.0f}"])
