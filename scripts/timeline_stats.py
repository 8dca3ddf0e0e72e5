"""Concurrency statistics from a rocprofv3 --kernel-trace database (rocpd sqlite): over the busiest window of the run
(AB: the last `frac` of the trace), how many train / reduce / apply kernels run at once, how long each queue idles
between consecutive kernels, and how long kernels take.  usage: timeline_stats.py run.db [frac=0.5]"""
import sqlite3, sys, collections
db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f'select name, start, "end", {qcol or 0} from kernels order by start').fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[1] >= lo]
span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
kind = lambda n: "train" if "k_train" in n else "reduce" if "k_reduce" in n else "apply" if "k_apply" in n else "other"
ev = []
for n, s, e, q in rows:
    ev.append((s, 1, kind(n))); ev.append((e, -1, kind(n)))
ev.sort()
act = collections.Counter(); hist = collections.Counter(); last = ev[0][0]
busy_any = 0
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        hist[(act["train"], act["apply"] > 0)] += dt
        if sum(act.values()) > 0: busy_any += dt
    act[k] += d; last = t
tot = sum(hist.values())
print(f"window {span/1e3:.1f} ms, kernels {len(rows)}; some kernel running {100*busy_any/tot:.1f} % of the time")
for (nt, ap), v in sorted(hist.items()):
    if v / tot > 0.005: print(f"  train kernels active {nt}, apply active {ap}: {100*v/tot:5.1f} %")
byq = collections.defaultdict(list)
for n, s, e, q in rows: byq[q].append((s, e, n))
for q, lst in byq.items():
    gaps = [lst[i+1][0] - lst[i][1] for i in range(len(lst)-1)]
    gaps = [g for g in gaps if g < 200000]
    if len(gaps) > 100:
        gaps.sort()
        print(f"  queue {q}: {len(lst)} kernels, gap to next kernel median {gaps[len(gaps)//2]/1e3:.2f} us, mean {sum(gaps)/len(gaps)/1e3:.2f} us, p90 {gaps[int(.9*len(gaps))]/1e3:.2f} us")
dur = collections.defaultdict(list)
for n, s, e, q in rows: dur[kind(n)].append(e - s)
for k, v in dur.items():
    v.sort(); print(f"  {k}: {len(v)} calls, duration median {v[len(v)//2]/1e3:.2f} us, mean {sum(v)/len(v)/1e3:.2f} us, total {sum(v)/1e6:.1f} ms")
