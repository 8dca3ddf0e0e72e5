"""gpurun_out/prof_TAG/*.csv (scripts/profile_round.sh) -> profiles/TAG_*.csv + profiles/pmc_summary.json, the file
bench.py reads `roofline.traffic` and `roofline.mfma_busy_frac` from.   python scripts/make_pmc_summary.py TAG"""
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
    with open(os.path.join(src, name)) as f:
        for r in csv.DictReader(f):
            out[(r["kernel"], r["counter"])] = (int(r["dispatches"]), float(r["mean_value"]), float(r["mean_duration_ns"]))
    return out


def pick(table, kernel_sub, counter):
    for (k, c), v in table.items():
        if kernel_sub in k and c == counter:
            return v
    return None


for f in ("kernel_stats_one_in_flight.csv", "kernel_stats_four_in_flight.csv", "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv",
          "sq_a.csv", "sq_b.csv", "bench_one_in_flight.json", "bench_four_in_flight.json", "kernel_stats_bc256.csv",
          "kernel_stats_embed.csv", "bench_bc256.json", "bench_embed.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
fetch, write = load("pmc_FETCH_SIZE.csv"), load("pmc_WRITE_SIZE.csv")
sqa, sqb = load("sq_a.csv"), load("sq_b.csv")
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes of `python3 scripts/prof_fit.py 2048 64 4` "
                 f"(one fit at a time, 4 epochs x 2 fits: scripts/profile_round.sh {tag}); per-kernel means in profiles/{tag}_pmc_*.csv and "
                 f"profiles/{tag}_sq_*.csv.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced read), "
                 "WRITE_SIZE is as read; both are KiB per launch"}
for key, sub in (("k_train_wave", "k_train_wave"), ("k_reduce_adam", "k_reduce_adam"), ("k_apply_mfma_eval", "k_apply_mfma<2, 1>")):
    f, w = pick(fetch, sub, "FETCH_SIZE"), pick(write, sub, "WRITE_SIZE")
    if f and w:
        out[f"{key}_fetch_size_avg_KB"] = f[1]
        out[f"{key}_write_size_avg_KB"] = w[1]
        out[f"{key}_hbm_bytes_per_launch"] = int((2 * f[1] + w[1]) * 1024)
    busy, insts = pick(sqa, sub, "SQ_VALU_MFMA_BUSY_CYCLES"), pick(sqb, sub, "SQ_INSTS_MFMA")
    if busy:
        dur_cycles = busy[2] * 1e-9 * 2.3e9          # kernel duration in shader cycles at ~2.3 GHz (in-kernel clock, stamps)
        out[f"{key}_mfma_busy_cycles_per_launch"] = busy[1]
        out[f"{key}_mfma_insts_per_launch"] = insts[1] if insts else None
        out[f"{key}_duration_us_in_counter_pass"] = round(busy[2] / 1e3, 2)
        out[f"{key}_mfma_busy_frac_whole_chip"] = round(busy[1] / (1024 * dur_cycles), 4)     # 256 CUs x 4 SIMDs
    wc, wa, wi, ac = (pick(sqb, sub, c) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
    if wc:
        out[f"{key}_wave_time_split"] = {"parked (waitcnt/barrier)": round(wa[1] / wc[1], 3), "issue-stalled": round(wi[1] / wc[1], 3),
                                         "issuing": round(ac[1] / wc[1], 3)}
out["k_train_wave_simds_occupied"] = 512
if "k_train_wave_mfma_busy_frac_whole_chip" in out:
    out["k_train_wave_mfma_busy_frac_occupied_simds"] = round(2 * out["k_train_wave_mfma_busy_frac_whole_chip"], 4)
out["algorithmic_bytes_per_train_launch"] = 8192 * 16
out["note"] = ("k_train_wave: 128 workgroups of 64 rows on 128 CUs.  Traffic per launch = the 9.5 MB gradient slabs written through (128 x 18,688 "
               "floats) + the 6.8 MB row gather (8192 x 832 B from the materialised row matrix) + the weights once per XCD, against 131 KB of "
               "algorithmic bytes: not HBM-bound (%.1f MB / %.1f us = %.2f TB/s); round 1's kernel moved 27.4 MB (256 slabs + the same rows)"
               % (out.get("k_train_wave_hbm_bytes_per_launch", 0) / 1e6, out.get("k_train_wave_duration_us_in_counter_pass", 1.0),
                  out.get("k_train_wave_hbm_bytes_per_launch", 0) / 1e6 / max(out.get("k_train_wave_duration_us_in_counter_pass", 1.0), 1e-9)))
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
