"""gpurun_out/prof_TAG/*.csv (scripts/profile_round.sh) -> profiles/TAG_*.csv + profiles/pmc_summary.json, the file
bench.py reads `roofline.traffic` and `roofline.mfma_busy_frac` from -- keyed by configuration (bc64 = BASELINE.json
configs[1], bc256 = configs[2], embed = configs[4]) and kernel.   python scripts/make_pmc_summary.py TAG"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def load(name):
    out = {}
    path = os.path.join(src, name)
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for r in csv.DictReader(f):
            out[(r["kernel"], r["counter"])] = (int(r["dispatches"]), float(r["mean_value"]), float(r["mean_duration_ns"]))
    return out


def pick(table, kernel_sub, counter):
    for (k, c), v in table.items():
        if kernel_sub in k and c == counter:
            return k, v
    return None, None


for f in sorted(os.listdir(src)):
    if f.endswith((".csv", ".json")) and not f.startswith("."):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
# (apply_eval: the evaluation pass as a fit runs it -- MODE_EVAL_FAST = 2; the canonical MODE_EVAL = 1 pass where a run used it)
KERNELS = {"bc64": {"train": "k_train_stream", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2>", "apply_decode": "k_apply_mfma<2, 0>"},
           "bc256": {"train": "k_train_wide", "reduce": "k_reduce_adam", "apply_eval": "k_apply_wide<16, 2, 2>", "apply_decode": "k_apply_wide<16, 2, 0>"},
           "embed": {"train": "k_train_stream", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2>", "apply_decode": "k_apply_mfma<2, 0>"},
           # the launches of a PAIR of bc64 fits stepping side by side (scripts/prof_pair.py): 2 x 128 workgroups, every CU
           "pair": {"train": "k_train_stream", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2>"}}
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes of scripts/prof_fit.py per "
                 f"configuration (one fit at a time: scripts/profile_round.sh {tag}); per-kernel means in profiles/{tag}_pmc_*.csv and "
                 f"profiles/{tag}_sq_*.csv.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced read), "
                 "WRITE_SIZE is as read; both KiB per launch",
       "configs": {}}
for cfg, kernels in KERNELS.items():
    fetch, write = load(f"pmc_{cfg}_FETCH_SIZE.csv"), load(f"pmc_{cfg}_WRITE_SIZE.csv")
    sqa, sqb = load(f"sq_a_{cfg}.csv"), load(f"sq_b_{cfg}.csv")
    cout = {}
    for key, sub in kernels.items():
        e = {}
        name, f = pick(fetch, sub, "FETCH_SIZE")
        _, w = pick(write, sub, "WRITE_SIZE")
        if f and w:
            e.update(kernel=name, fetch_size_avg_KB=f[1], write_size_avg_KB=w[1], hbm_bytes_per_launch=int((2 * f[1] + w[1]) * 1024))
        _, busy = pick(sqa, sub, "SQ_VALU_MFMA_BUSY_CYCLES")
        _, insts = pick(sqb, sub, "SQ_INSTS_MFMA")
        _, conf = pick(sqa, sub, "SQ_LDS_BANK_CONFLICT")
        _, ldsact = pick(sqa, sub, "SQ_ACTIVE_INST_LDS")
        _, coexec = pick(sqb, sub, "SQ_VALU_MFMA_COEXEC_CYCLES")
        if busy:
            dur_cycles = busy[2] * 1e-9 * 2.3e9          # kernel duration in shader cycles at ~2.3 GHz (in-kernel clock, stamps)
            e.update(mfma_busy_cycles_per_launch=busy[1], mfma_insts_per_launch=insts[1] if insts else None,
                     duration_us_in_counter_pass=round(busy[2] / 1e3, 2),
                     mfma_busy_frac_whole_chip=round(busy[1] / (1024 * dur_cycles), 4))     # 256 CUs x 4 SIMDs
            if key == "train":   # 128 workgroups (pair: 256), one compute wave per SIMD: 512 (1024) of the chip's 1024 SIMDs
                e["mfma_busy_frac_occupied_simds"] = round((1 if cfg == "pair" else 2) * e["mfma_busy_frac_whole_chip"], 4)
        if conf:
            e.update(lds_bank_conflict_cycles_per_launch=conf[1], lds_active_cycles_per_launch=ldsact[1] if ldsact else None)
        if coexec:
            e["valu_mfma_coexec_cycles_per_launch"] = coexec[1]
        wc, wa, wi, ac = (pick(sqb, sub, c)[1] for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
        if wc:
            e["wave_time_split"] = {"parked (waitcnt/barrier)": round(wa[1] / wc[1], 3), "issue-stalled": round(wi[1] / wc[1], 3),
                                    "issuing": round(ac[1] / wc[1], 3)}
        if e:
            cout[key] = e
    if cout:
        cout["algorithmic_bytes_per_train_launch"] = 8192 * 16 * (2 if cfg == "pair" else 1)
        out["configs"][cfg] = cout
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
