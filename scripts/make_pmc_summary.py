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
KERNELS = {"bc64": {"train": "k_train_stream<48", "train_split": "k_train_split<48", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2", "apply_decode": "k_apply_mfma<2, 0",
                    "build_rows": "k_build_rows_tiled"},
           "bc256": {"train": "k_train_half", "dw": "k_dw_wide", "reduce": "k_reduce_adam", "apply_eval": "k_apply_wide<16, 2, 2>", "apply_decode": "k_apply_wide<16, 2, 0>"},
           "embed": {"train": "k_train_stream<64", "train_split": "k_train_split<64", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2", "apply_decode": "k_apply_mfma<2, 0"},
           # the launches of a PAIR of bc64 fits stepping side by side (scripts/prof_pair.py): 2 x 128 workgroups, every CU
           "pair": {"train": "k_train_stream<48", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2"},
           # the reference's 4-band shape (F = 100: run.sh:14-28), one fit alone and the pair launch
           "bands4": {"train": "k_train_stream<24", "train_split": "k_train_split<24", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2", "apply_decode": "k_apply_mfma<2, 0",
                      "build_rows": "k_build_rows_tiled"},
           "bands4_pair": {"train": "k_train_stream<24", "reduce": "k_reduce_adam", "apply_eval": "k_apply_mfma<2, 2"}}
# MI355X_MICROARCH.md, HBM: "on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B per
# lane, global_load and buffer_load ... lds alike)" -- so the doubling applies to the kernels whose fetches ARE such reads
# (rows by LDS-DMA, fragments / weights / slabs as 16-byte loads) and NOT to the apply kernels, which fetch two uint16 planes
# in 2- and 4-byte pieces: their un-doubled 134.0 MB per pass is exactly img + msb
FETCH_DOUBLED = ("k_train_stream", "k_train_split", "k_train_wide", "k_train_half", "k_reduce_adam", "k_dw_wide", "k_train_mfma")
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes of scripts/prof_fit.py per "
                 f"configuration (one fit at a time: scripts/profile_round.sh {tag}); per-kernel means in profiles/{tag}_pmc_*.csv and "
                 f"profiles/{tag}_sq_*.csv.  FETCH_SIZE is doubled for the kernels whose fetches are 16-byte-per-lane streaming reads "
                 f"({', '.join(FETCH_DOUBLED)}: MI355X_MICROARCH.md, gfx950 reports half of those) and taken as read for the others "
                 "(`fetch_doubled` per kernel); WRITE_SIZE is as read; both KiB per launch.  kernel_trace: rocprofv3 --kernel-trace "
                 f"--stats averages of the training launch (profiles/{tag}_kernel_stats_*.csv); timeline / stamps: in-kernel "
                 f"s_memrealtime / s_memtime summaries of the diagnostic builds (scripts/collect_inkernel.py, profiles/{tag}_timeline_*.json, "
                 f"{tag}_stamps_*.json)",
       "configs": {}}


def trace_avg(csv_name, kernel_sub):
    """(average us, minimum us, calls, full kernel name) of the first kernel whose name contains kernel_sub"""
    path = os.path.join(src, csv_name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_sub in r["Name"]:
                return round(float(r["AverageNs"]) / 1e3, 2), round(float(r["MinNs"]) / 1e3, 2), int(r["Calls"]), r["Name"]
    return None


def load_json(name):
    path = os.path.join(src, name)
    return json.load(open(path)) if os.path.exists(path) else None


# configuration -> the kernel-trace summary in which its training launch runs ALONE on the device (one chain)
TRACE = {"bc64": "kernel_stats_one_in_flight.csv", "pair": "kernel_stats_pair_alone.csv", "bc256": "kernel_stats_alone_bc256.csv",
         "embed": "kernel_stats_alone_embed.csv", "embed_pair": "kernel_stats_pair_alone_embed.csv",
         "bands4": "kernel_stats_alone_bands4.csv", "bands4_pair": "kernel_stats_pair_alone_bands4.csv"}
# ... and the summary of the TIMED configuration (bench.py's default run: two chains of pair launches in flight), whose
# average of the same kernel is longer -- the chains take turns on a chip either of them fills
TRACE_TIMED = {"pair": "kernel_stats_four_in_flight.csv", "embed_pair": "kernel_stats_embed.csv", "bands4_pair": "kernel_stats_bands4.csv",
               "bc256": "kernel_stats_bc256.csv"}
for cfg, kernels in KERNELS.items():
    fetch, write = load(f"pmc_{cfg}_FETCH_SIZE.csv"), load(f"pmc_{cfg}_WRITE_SIZE.csv")
    sqa, sqb = load(f"sq_a_{cfg}.csv"), load(f"sq_b_{cfg}.csv")
    cout = {}
    for key, sub in kernels.items():
        e = {}
        name, f = pick(fetch, sub, "FETCH_SIZE")
        _, w = pick(write, sub, "WRITE_SIZE")
        if f and w:
            dbl = any(k in name for k in FETCH_DOUBLED)
            e.update(kernel=name, fetch_size_avg_KB=f[1], write_size_avg_KB=w[1], fetch_doubled=dbl,
                     hbm_bytes_per_launch=int(((2 if dbl else 1) * f[1] + w[1]) * 1024))
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
            if key == "train_split":   # 256 workgroups of four compute waves: every SIMD of the chip
                e["mfma_busy_frac_occupied_simds"] = e["mfma_busy_frac_whole_chip"]
            if key == "train":   # 128 workgroups of one fit (pair, and the 256 32-row workgroups of bc256: every CU), one compute
                                 # wave per SIMD: 512 (1024) of the chip's 1024 SIMDs
                e["mfma_busy_frac_occupied_simds"] = round((1 if (cfg in ("pair", "bc256") or cfg.endswith("_pair")) else 2) * e["mfma_busy_frac_whole_chip"], 4)
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
        cout["algorithmic_bytes_per_train_launch"] = 8192 * (8 if cfg.startswith("bands4") else 16) * (2 if (cfg == "pair" or cfg.endswith("_pair")) else 1)
        out["configs"][cfg] = cout
for cfg, csv_name in list(TRACE.items()) + [("split", None), ("split_embed", None)]:
    t = trace_avg(csv_name, KERNELS.get(cfg, KERNELS["embed"])["train"]) if csv_name else None
    cout = out["configs"].setdefault(cfg, {})
    if t:
        cout["kernel_trace"] = {"train_avg_us": t[0], "train_min_us": t[1], "train_calls": t[2], "kernel": t[3],
                                "source": f"profiles/{tag}_{csv_name}"}
        r = trace_avg(csv_name, "k_reduce_adam")
        if r:
            cout["kernel_trace"].update(reduce_avg_us=r[0], reduce_min_us=r[1])
        d = trace_avg(csv_name, "k_dw_wide") if cfg == "bc256" else None
        if d:
            cout["kernel_trace"].update(dw_avg_us=d[0], dw_min_us=d[1])
        tt = trace_avg(TRACE_TIMED[cfg], KERNELS.get(cfg, KERNELS["embed"])["train"]) if cfg in TRACE_TIMED else None
        if tt:
            cout["kernel_trace"].update(timed_region_avg_us=tt[0], timed_region_min_us=tt[1], timed_region_calls=tt[2],
                                        timed_region_source=f"profiles/{tag}_{TRACE_TIMED[cfg]}")
            if cfg == "bc256":
                td = trace_avg(TRACE_TIMED[cfg], "k_dw_wide")
                if td:
                    cout["kernel_trace"].update(timed_region_dw_avg_us=td[0])
        sp = trace_avg(csv_name, KERNELS.get(cfg, {}).get("train_split", "k_train_split<none"))
        if sp:   # the every-CU launch of a lone fit (most of the steps of such a trace; the rest go out on the half-chip launch above)
            cout["kernel_trace"].update(split_avg_us=sp[0], split_min_us=sp[1], split_calls=sp[2], split_kernel=sp[3])
    for kind in ("timeline", "stamps"):
        j = load_json(f"{kind}_{cfg}.json")
        if j and (j.get("per_step_us") or j.get("mean_cycles_per_phase")):
            j["source"] = f"profiles/{tag}_{kind}_{cfg}.json"
            cout[kind] = {k: j[k] for k in ("what", "per_step_us", "mean_cycles_per_phase", "clock_MHz", "wave_lifetime_cycles",
                                            "first_start_to_last_end_us", "source") if k in j}
    if not cout:
        del out["configs"][cfg]
for extra in ("timeline_bc64_two_chains.json", "timeline_pair_two_chains.json", "timeline_bands4_pair_two_chains.json"):   # the timed region itself: two chains in flight
    j = load_json(extra)
    if j and j.get("per_step_us"):
        out.setdefault("timeline_in_flight", {})[extra[len("timeline_"):-len(".json")]] = {
            "what": j["what"], "per_step_us": j["per_step_us"], "source": f"profiles/{tag}_{extra}"}
# a partial re-collection (e.g. kernel traces and in-kernel summaries only) keeps the counter entries of the previous
# summary, each marked with where it came from
old_path = os.path.join(dst, "pmc_summary.json")
if os.path.exists(old_path):
    old = json.load(open(old_path))
    for cfg, oc in old.get("configs", {}).items():
        nc = out["configs"].setdefault(cfg, {})
        for k, v in oc.items():
            if k not in nc:
                if isinstance(v, dict) and "carried_over_from" not in v:
                    v = dict(v, carried_over_from=old.get("source", "an earlier summary")[:160])
                nc[k] = v
json.dump(out, open(old_path, "w"), indent=1)
print(json.dumps(out, indent=1))
