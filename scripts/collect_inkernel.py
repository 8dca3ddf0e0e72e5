"""In-kernel evidence as DATA: runs the stamp / timeline probes against the diagnostic builds of
the library (lbdrn-msic_amd/liblbdrn_hip_stamps.so: -DLBDRN_TRAIN_STAMPS -DLBDRN_APPLY_STAMPS, per-phase s_memtime
stamps of every compute wave; liblbdrn_hip_timeline.so: -DLBDRN_TIMELINE, s_memrealtime of the first wave's start and
the last store of both launches of every step), parses what the library prints and writes one JSON per configuration:

    gpurun_out/prof_TAG/stamps_<config>.json     mean cycles per phase of a compute wave (the DESIGN.md 4.1 breakdown)
    gpurun_out/prof_TAG/timeline_<config>.json   training launch | gap | reduce launch | gap, microseconds per step

scripts/make_pmc_summary.py copies them to profiles/TAG_* and folds them into profiles/pmc_summary.json, from where
bench.py puts them into the BENCH line.  Run ON the GPU box:   python3 scripts/collect_inkernel.py TAG [config ...]
(build the two libraries first: python lbdrn-msic_amd/csrc/build.py --stamps; ... --variant timeline -DLBDRN_TIMELINE)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "lbdrn-msic_amd")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
OUT = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
os.makedirs(OUT, exist_ok=True)

# configuration -> (what it is, probe command).  bc64 = one fit per launch alone on the device (BASELINE configs[1] as
# a single tile runs it), pair = two fits per launch (the launch of the timed region), pair2 = two such chains in
# flight (the timed region itself), bc256 = configs[2], embed = configs[4] (one fit per launch)
STAMPS = {
    "bc64": ("one fit per launch, alone on the device", ["scripts/stamp_probe_group.py", "1"]),
    "pair": ("two fits per launch (2 x 128 workgroups), alone on the device", ["scripts/stamp_probe_group.py", "2"]),
    "bc256": ("bc = 256, one fit per launch, alone on the device", ["scripts/stamp_probe_wide.py", "2048"]),
    "embed": ("USE_COORDINATES + EMBEDDING, one fit per launch, alone", ["scripts/stamp_probe_group.py", "1", "embed"]),
    # the every-CU launch of a fit that has the device to itself (k_train_split: 256 workgroups of 32 rows), real epochs
    "split": ("one fit per launch on every CU (LBDRN_TRAIN_ALONE), alone on the device", ["scripts/lone_step_probe.py", "3", "--only", "split", "--modes", "0"]),
    "split_embed": ("USE_COORDINATES + EMBEDDING, every-CU launch, alone", ["scripts/lone_step_probe.py", "3", "--embed", "--only", "split", "--modes", "0"]),
    # the reference's 4-band shape (F = 100): the pair launch of fits in flight and the every-CU launch of a lone fit
    "bands4_pair": ("4 bands, two fits per launch (2 x 128 workgroups), alone on the device", ["scripts/stamp_probe_group.py", "2", "bands4"]),
    "bands4": ("4 bands, one fit per launch on every CU (LBDRN_TRAIN_ALONE), alone on the device", ["scripts/lone_step_probe.py", "3", "--bands4", "--only", "split", "--modes", "0"]),
}
TIMELINE = {
    "bc64": ("one chain of single-fit launches, alone on the device", ["scripts/stamp_probe_inflight.py", "1", "3", "8", "2", "1"]),
    "bc64_two_chains": ("two chains of single-fit launches in flight", ["scripts/stamp_probe_inflight.py", "2", "3", "8", "2", "1"]),
    "pair": ("one chain of pair launches, alone on the device", ["scripts/stamp_probe_inflight.py", "2", "3", "8", "2", "2"]),
    "pair_two_chains": ("two chains of pair launches in flight (the timed region of bench.py)",
                        ["scripts/stamp_probe_inflight.py", "4", "3", "8", "2", "2"]),
    "split": ("one chain of every-CU launches (k_train_split) and their reduce launches over slab pairs, alone on the device",
              ["scripts/lone_step_probe.py", "3", "--only", "split", "--modes", "0"]),
    "bands4_pair": ("4 bands: one chain of pair launches, alone on the device", ["scripts/stamp_probe_inflight.py", "2", "3", "4", "2", "2"]),
    "bands4_pair_two_chains": ("4 bands: two chains of pair launches in flight (the bands4 leg of bench.py)",
                               ["scripts/stamp_probe_inflight.py", "4", "3", "4", "2", "2"]),
    "bands4": ("4 bands: one chain of every-CU launches (k_train_split<24,6>), alone on the device",
               ["scripts/lone_step_probe.py", "3", "--bands4", "--only", "split", "--modes", "0"]),
}


def run(lib, cmd):
    env = dict(os.environ, LBDRN_HIP_LIB=os.path.join(LIBDIR, lib), GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "8"))
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    return r.returncode, r.stderr


def pieces(line, after):
    """'... after: label 12 | label 34 us | ...' -> [(label, value)]"""
    body = line.split(after, 1)[1]
    out = []
    for part in body.split("|"):
        m = re.match(r"\s*(.*?)\s*(-?\d+(?:\.\d+)?)\s*(us)?\s*$", part.strip())
        if m:
            out.append((m.group(1).strip(" ->") or "step", float(m.group(2))))
    return out


def mean_rows(rows):
    keys = [k for k, _ in rows[0]]
    return {k: round(sum(dict(r)[k] for r in rows) / len(rows), 2) for k in keys}


def collect_stamps(name, what, cmd):
    rc, err = run("liblbdrn_hip_stamps.so", cmd)
    lines = [l for l in err.splitlines() if l.startswith("[lbdrn stamps")]
    res = {"config": name, "what": what, "command": "LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_stamps.so python3 " + " ".join(cmd),
           "rc": rc, "epochs_printed": len(lines)}
    if lines:
        use = lines[1:] or lines          # the first epoch call of a process starts cold
        res["kernel"] = re.match(r"\[lbdrn stamps, ([^\]]*)\]", use[0]).group(1) if "," in use[0].split("]")[0] else "tile kernel"
        head = use[-1].split("mean cycles:")[0]
        for key, pat in (("clock_MHz", r"clock (\d+) MHz"), ("wave_lifetime_cycles", r"wave lifetime (\d+) cycles"),
                         ("first_start_to_last_end_us", r"first start -> last end ([\d.]+) us")):
            vals = [float(m.group(1)) for l in use for m in [re.search(pat, l)] if m]
            if vals:
                res[key] = round(sum(vals) / len(vals), 2)
        res["mean_cycles_per_phase"] = mean_rows([pieces(l, "mean cycles:") for l in use])
        res["phases_sum_cycles"] = round(sum(res["mean_cycles_per_phase"].values()), 1)
        res["note"] = ("mean over the compute waves of the LAST training launch of each epoch call, averaged over "
                       f"{len(use)} epoch calls (the first call of the process left out); cycles of the shader clock "
                       "(s_memtime), clock from s_memrealtime over the wave's lifetime")
    else:
        res["stderr_tail"] = err[-1500:]
    json.dump(res, open(os.path.join(OUT, f"stamps_{name}.json"), "w"), indent=1)
    print(json.dumps(res))


def collect_timeline(name, what, cmd):
    rc, err = run("liblbdrn_hip_timeline.so", cmd)
    lines = [l for l in err.splitlines() if l.startswith("[lbdrn timeline]")]
    res = {"config": name, "what": what, "command": "LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_timeline.so python3 " + " ".join(cmd),
           "rc": rc, "epochs_printed": len(lines)}
    if lines:
        n_chains = 1 if "lone_step_probe" in cmd[0] else max(1, int(cmd[1]) // max(1, int(cmd[5]) if len(cmd) > 5 else 1))
        use = lines[n_chains:] or lines    # the first epoch of every chain starts behind the permutation pipeline
        rows = [pieces(l, "steps:") for l in use]
        names = ["train_first_wave_start_to_last_store_us", "gap_to_reduce_start_us", "reduce_us", "gap_to_next_train_us", "step_us"]
        rows = [list(zip(names, [v for _, v in r])) for r in rows]
        res["per_step_us"] = mean_rows(rows)
        res["note"] = (f"s_memrealtime (100 MHz) stamps of workgroup 0's first wave and of every wave's last store, per step, "
                       f"steps 8.. of each epoch; mean over {len(use)} epoch calls of all chains in flight")
    else:
        res["stderr_tail"] = err[-1500:]
    json.dump(res, open(os.path.join(OUT, f"timeline_{name}.json"), "w"), indent=1)
    print(json.dumps(res))


want = sys.argv[2:]
for name, (what, cmd) in STAMPS.items():
    if not want or f"stamps_{name}" in want or "stamps" in want:
        collect_stamps(name, what, cmd)
for name, (what, cmd) in TIMELINE.items():
    if not want or f"timeline_{name}" in want or "timeline" in want:
        collect_timeline(name, what, cmd)
