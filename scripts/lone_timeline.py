"""Where the wall time of ONE fit alone on the device goes, from a rocprofv3 --kernel-trace database (rocpd sqlite) of
scripts/prof_fit.py: the step chain (k_train / k_reduce) of the LAST fit in the trace -- time inside its kernels, the
short boundaries between them, and every longer hole in the chain with the kernels that ran meanwhile.
usage: lone_timeline.py run.db"""
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute('select name, start, "end" from kernels order by start').fetchall()
short = lambda n: n.split("(")[0].replace("void lbdrn::", "").replace("lbdrn::", "")[:40]
chain = [(s, e, n) for n, s, e in rows if "k_train" in n or "k_reduce" in n]
# the fits of the trace: a hole of more than 3 ms in the chain separates them
fits, cur_fit = [], [chain[0]]
for a, b in zip(chain, chain[1:]):
    if b[0] - a[1] > 3e6 and len(cur_fit) > 5000: fits.append(cur_fit); cur_fit = []
    cur_fit.append(b)
fits.append(cur_fit)
fit = fits[-1]
t0, t1 = fit[0][0], fit[-1][1]
# everything from the kernel before the chain that is not part of the previous fit to the last kernel after it
prev_end = fits[-2][-1][1] if len(fits) > 1 else rows[0][1]
others = [(s, e, n) for n, s, e in rows if s >= prev_end and not ("k_train" in n or "k_reduce" in n)]
first = min([s for s, e, n in others] + [t0])
last = max([e for s, e, n in others] + [t1])
print(f"fits in the trace: {len(fits)}; last fit: {len(fit)} chain kernels")
print(f"first kernel of the fit -> first training launch {(t0 - first)/1e6:.2f} ms; chain {(t1 - t0)/1e6:.2f} ms; "
      f"last reduce -> last kernel {(last - t1)/1e6:.2f} ms; total {(last - first)/1e6:.2f} ms")
inside = collections.Counter(); ninside = collections.Counter()
for s, e, n in fit: inside[short(n)] += e - s; ninside[short(n)] += 1
for k, v in inside.items(): print(f"  in {k}: {v/1e6:.2f} ms ({ninside[k]} launches, {v/ninside[k]/1e3:.2f} us each)")
gaps = [(b[0] - a[1], a[1], b[0], short(a[2]), short(b[2])) for a, b in zip(fit, fit[1:])]
small = [g for g in gaps if g[0] < 20e3]
big = [g for g in gaps if g[0] >= 20e3]
print(f"  boundaries < 20 us: {len(small)}, {sum(g[0] for g in small)/1e6:.2f} ms (mean {sum(g[0] for g in small)/len(small)/1e3:.2f} us)")
print(f"  holes >= 20 us: {len(big)}, {sum(g[0] for g in big)/1e6:.2f} ms")
for g in sorted(big, key=lambda g: g[1]):
    during = collections.Counter()
    for s, e, n in others:
        ov = min(e, g[2]) - max(s, g[1])
        if ov > 0: during[short(n)] += ov
    txt = ", ".join(f"{k} {v/1e3:.0f} us" for k, v in during.most_common(4))
    print(f"    at {(g[1]-t0)/1e6:7.2f} ms: {g[0]/1e3:7.1f} us after {g[3]} before {g[4]}; meanwhile: {txt}")
# how the training kernels stretch while an evaluation pass runs beside them
apply_iv = [(s, e) for s, e, n in others if "k_apply" in n]
def beside(s, e): return any(min(e, b) - max(s, a) > 0 for a, b in apply_iv)
for kind in ("k_train", "k_reduce"):
    a = [e - s for s, e, n in fit if kind in n and beside(s, e)]
    b = [e - s for s, e, n in fit if kind in n and not beside(s, e)]
    if a and b: print(f"  {kind}: {len(a)} launches beside an evaluation pass {sum(a)/len(a)/1e3:.2f} us, {len(b)} without {sum(b)/len(b)/1e3:.2f} us")
