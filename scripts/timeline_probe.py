"""From a rocprofv3 rocpd database: for the LAST fit in the trace, when do the evaluation kernels run relative to the
train kernels (per queue)?   python scripts/timeline_probe.py run.db"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
sel = f'select name, start, "end", {qcol if qcol else 0}, {"stream_id" if "stream_id" in cols else 0} from kernels order by start'
rows = cur.execute(sel).fetchall()
t_end = rows[-1][2]
last = [r for r in rows if r[1] > t_end - float(sys.argv[2] if len(sys.argv) > 2 else 1.3) * 1e9]
t0 = last[0][1]
ev = [r for r in last if "k_apply" in r[0]]
tr = [r for r in last if "k_train" in r[0]]
print("train kernels", len(tr), "queues", sorted({r[3] for r in tr}), "streams", sorted({r[4] for r in tr}))
print("apply kernels", len(ev), "queues", sorted({r[3] for r in ev}), "streams", sorted({r[4] for r in ev}))
for r in ev[-12:]:
    inside = [t for t in tr if t[1] >= r[1] and t[2] <= r[2]]
    mean = sum(t[2] - t[1] for t in inside) / max(len(inside), 1)
    print(f"{r[0][:40]:40s} q{r[3]} start {(r[1]-t0)/1e6:9.2f} ms  dur {(r[2]-r[1])/1e6:7.2f} ms   train kernels inside: {len(inside):4d}  mean {mean/1e3:7.1f} us")
out = [t for t in tr if not any(t[1] >= r[1] and t[2] <= r[2] for r in ev)]
print("train kernels outside any apply kernel:", len(out), "mean", sum(t[2]-t[1] for t in out)/max(len(out),1)/1e3, "us")

tr.sort(key=lambda r: r[1])
gaps = [tr[k + 1][1] - tr[k][2] for k in range(len(tr) - 1)]
gaps.sort()
print("train kernel duration mean %.1f us; gap to the next train kernel: median %.1f us, p90 %.1f us" % (
    sum(t[2] - t[1] for t in tr) / len(tr) / 1e3, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3))
names = {}
for r in last:
    names.setdefault((r[0][:50], r[3]), [0, 0]); names[(r[0][:50], r[3])][0] += 1; names[(r[0][:50], r[3])][1] += r[2] - r[1]
for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k[0]:50s} q{k[1]}  {v[0]:6d} calls  total {v[1]/1e6:8.2f} ms  mean {v[1]/v[0]/1e3:8.1f} us")
