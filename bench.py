#!/usr/bin/env python3
"""Benchmark of the LBDRN per-image encode+decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU, over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic tile: the fused fit (bit split, 10 epochs of
minibatch Adam with a whole-image evaluation after each, best-epoch selection), the 16-bit weight
truncation the bitstream applies, and the fused apply (reconstruction).  The K tiles of the timed region are
independent fits; --in-flight of them (default 4) progress at a time on each GPU, on their own streams: one
fit's training step occupies half the chip (128 workgroups), so two steps of different fits run side by side while the
reduce / evaluation / permutation kernels of the others fill the rest.  Each rank owns its own
tiles (images are independent fits): weak scaling, no data-path collective; RCCL
carries only the max-over-ranks time and the per-image metric records.  Inputs are resident in HBM
when the timed region starts; host work that belongs to the path (the permutations of
torch.randperm order) is inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
# the fits in flight, the permutation side stream and torch's own streams should each get a hardware queue of
# their own (the runtime's default is four per process; streams that share one run one after the other)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SEED = 19920517  # ref encode.py:169


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None,
                   help="timed tiles per GPU (default: 16 on one GPU -- four rounds of the four fits in flight, the first of "
                        "which start in lockstep -- and 8 with --gpus N > 1: at 8 GPUs the 64-tile job of BASELINE.json "
                        "configs[3])")
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--height", type=int, default=2048)
    p.add_argument("--width", type=int, default=2048)
    p.add_argument("--bands", type=int, default=8)
    p.add_argument("-K", type=int, default=5)
    p.add_argument("-D", type=int, default=2)
    p.add_argument("-bc", type=int, default=64)
    p.add_argument("-nl", type=int, default=2)
    p.add_argument("-bs", type=int, default=8192)
    p.add_argument("-e", "--epochs", type=int, default=10)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--path", choices=["auto", "generic", "mfma"], default="auto")
    p.add_argument("--coords-embedding", action="store_true",
                   help="BASELINE.json configs[4]: USE_COORDINATES=True + EMBEDDING=True (F = 250)")
    p.add_argument("--in-flight", type=int, default=4,
                   help="tiles progressing at a time on each GPU (independent fits on their own streams)")
    p.add_argument("--repeats", type=int, default=3, help="repeats of the timed region; the median is reported")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip the short legs of BASELINE.json configs[2] (bc = 256) and configs[4] (coordinates + embedding) "
                        "that a default one-GPU run of the headline configuration appends as `other_configs`")
    p.add_argument("--cpu-sample", type=int, default=512, help="side of the CPU-baseline crop")
    p.add_argument("--cpu-full-epochs", type=int, default=10,
                   help="epochs of form B (vectorised) measured on the WHOLE tile, scaled to the full recipe (0: skip)")
    p.add_argument("--cpu-epochs", type=int, default=1,
                   help="epochs of the recipe the CPU baseline runs on its crop (scaled to the full recipe)")
    a = p.parse_args()
    if a.steps is None:
        a.steps = 16 if a.gpus == 1 else 8
    return a


def feat_cfg(a):
    from lbdrn_hip.features import FeatCfg
    return FeatCfg(use_coordinates=a.coords_embedding, embedding=a.coords_embedding)


def run_images(codec, ops, tiles, a, path):
    """encode (fit) + weight truncation + decode (apply) for each HBM-resident tile; a.in_flight tiles progress
    at a time.  Every fit seeds itself like an encode.py invocation does (ref encode.py:200-205)."""
    def finish(fit):
        params = codec.truncate_device(fit.best_params, 16)
        return fit, codec.apply_device(fit.geom, fit.net, fit.msb, params, path=path)
    return codec.fit_many(tiles, a.K, a.D, a.bc, a.nl, a.lr, a.bs, a.epochs, cfg=feat_cfg(a), path=path,
                          seed=SEED, in_flight=a.in_flight, then=finish)


def tiles_per_launch(a, ops, path):
    """How many fits one training launch of the timed region carries (codec.fit_many: pairs with four or more of one
    shape in flight on the streamed bc = 64 step)."""
    if min(a.in_flight, a.steps) < 4 or path == ops._lib.PATH_GENERIC:
        return 1
    n = int(os.environ.get("LBDRN_FIT_GROUP", "0")) or (
        2 if ops.train_group_size(a.bands, a.height, a.width, a.K, a.D, feat_cfg(a), a.bc, a.nl) >= 2 else 1)
    return max(1, min(n, ops.train_group_max()))


def flops_per_pixel(F, bc, C, nl):
    fwd = 2 * (F * bc + (nl - 1) * bc * bc + bc * C)
    bwd = fwd + 2 * ((nl - 1) * bc * bc + bc * C)   # dW (same as fwd) + dX of every layer but the first
    return fwd, fwd + bwd


def event_time_ms(fn, stream, repeat=1):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(stream)
    for _ in range(repeat):
        fn()
    e.record(stream)
    e.synchronize()
    return s.elapsed_time(e) / repeat


def config_key(a):
    """Which profiled configuration of profiles/pmc_summary.json this run is (None: not one of them)."""
    if a.bands == 8 and a.K == 5 and a.D == 2 and a.nl == 2:
        if a.coords_embedding and a.bc == 64:
            return "embed"
        if not a.coords_embedding and a.bc in (64, 256):
            return f"bc{a.bc}"
    if a.bands == 4 and a.K == 5 and a.D == 2 and a.nl == 2 and a.bc == 64 and not a.coords_embedding:
        return "bands4"     # the reference's majority shape (run.sh:14-28: 9 of 13 images have four bands; F = 100)
    return None


def committed_profile(key, fits_per_launch):
    """What profiles/pmc_summary.json (scripts/make_pmc_summary.py, committed) holds for this configuration: the counter
    pass, the rocprofv3 --kernel-trace average of the training launch, the in-kernel timeline and stamp summaries."""
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not (os.path.exists(pmc) and key):
        return {}, {}
    try:
        d = json.load(open(pmc))
    except Exception:
        return {}, {}
    cfgs = d.get("configs", {})
    if key == "bc64" and fits_per_launch == 2:
        return cfgs.get("pair", {}), d
    prof = dict(cfgs.get(key, {}))
    if fits_per_launch == 2 and cfgs.get(key + "_pair"):   # the pair launch's own trace and, where a counter pass of it exists, counters
        pc = cfgs[key + "_pair"]
        for k in ("kernel_trace", "train", "reduce", "apply_eval", "timeline", "stamps"):
            if pc.get(k):
                prof[k] = pc[k]
        prof["_pair_counters"] = bool(pc.get("train", {}).get("hbm_bytes_per_launch"))
    return prof, d


def hbm_fields(a, seconds_per_tile, key, fits_per_launch):
    """north_star asks for the fraction of the HBM roofline next to the matrix one.  Two figures per tile, both against
    8 TB/s: the ALGORITHMIC bytes (16 B a pixel and pass -- one read of the uint16 source -- x 2 e passes, + 32 for
    the split, + 32 for the decode = 384 B a pixel at e = 10: 1.61 GB per tile) and the bytes the hardware COUNTED
    (profiles/pmc_summary.json: FETCH_SIZE / WRITE_SIZE per launch of every kernel of the path, separate --pmc passes,
    x the launches a tile makes).  The second is ~90 x the first -- gradient slabs written through and read back, the
    materialised row matrix gathered 832 B a row -- and is what the path really moves; timing-only builds price it at
    <= 2.4 % of a tile: the path is matrix-bound, which is why `bound` says mfma."""
    px = a.height * a.width
    e = a.epochs
    alg = (16 * 2 * e + 32 + 32) * px * (a.bands / 8.0)
    out = {"hbm_algorithmic_bytes_per_tile": int(alg), "hbm_algorithmic_GBps": round(alg / seconds_per_tile / 1e9, 1),
           "hbm_frac_algorithmic": round(alg / seconds_per_tile / 8e12, 5), "hbm_peak_GBps": 8000.0}
    prof, whole = committed_profile(key if px == 2048 * 2048 and a.bands == 8 else None, fits_per_launch)
    if not prof:
        out["counted_bytes_per_tile"] = None
        return out
    steps = e * ((px + a.bs - 1) // a.bs)
    per_fit = 1.0 / max(fits_per_launch, 1)
    parts = {}
    for name, launches in (("train", steps * per_fit), ("dw", steps), ("reduce", steps * per_fit), ("apply_eval", e if e > 1 else 0),
                           ("apply_decode", 1), ("build_rows", 1)):
        b = prof.get(name, {}).get("hbm_bytes_per_launch")
        if name == "apply_decode" and b is None:     # (the pair configuration is counted on fits only: the decode pass of the one-fit run)
            b = whole.get("configs", {}).get(key, {}).get("apply_decode", {}).get("hbm_bytes_per_launch")
        if name == "build_rows" and b is None:       # (not in the counter passes: its algorithmic bytes -- the row matrix written once, both planes read)
            b = px * (832 if key != "embed" else 1088) + 2 * a.bands * px * 2 if key != "bc256" else px * 832 + 2 * a.bands * px * 2
        if b is not None and launches:
            parts[name] = int(b * launches)
    total = sum(parts.values())
    out.update({"counted_bytes_per_tile": total, "counted_bytes_by_kernel": parts,
                "counted_TBps": round(total / seconds_per_tile / 1e12, 3), "counted_frac_of_hbm_peak": round(total / seconds_per_tile / 8e12, 4),
                "counted_over_algorithmic": round(total / alg, 1),
                "counted_source": "profiles/pmc_summary.json x launches per tile (per-launch bytes of a launch of "
                                  f"{fits_per_launch} fit(s), shared between them); kernels without a counter pass at their algorithmic bytes"})
    return out


def roofline_probe(codec, ops, fit, img_d, a, path, per_launch=None, with_single=True):
    """Live HIP-event timing, on the launch stream, right after the timed region (same process, same tile).
    Dominant kernel = the fused training step (5120 launches per tile, ~55 % of the GPU time of a fit).

    `kernel_us` -- the figure `achieved` and `frac` are computed from -- is the training kernel's OWN average duration:
    one 512-launch epoch of training launches back to back between one event pair (lbdrn_train_profile_mode 3: the
    reduce/Adam launches left out, every launch on its own slice of the permutation = cold rows, its launch boundary
    included).  Beside it, from the same probe: the whole step (train, reduce/Adam, train, ...: `train_step_pair_us`) and
    what one MORE launch of either kernel costs inside that dependent sequence (`marginal_us`, `reduce_adam_us`: the
    epoch with every training / reduce launch doubled -- the doubled launch finds its rows warm, so this is a lower
    bound of the kernel, reported for the step's accounting only).  The committed rocprofv3 --kernel-trace average of the
    same launch sequence (profiles/, via pmc_summary.json) rides along as `rocprof_kernel_us`, and `frac` is the SMALLER
    of the two fractions, so that it never exceeds what the committed trace gives.  Counter-derived fields (traffic,
    mfma_busy_frac) come from the committed rocprofv3 --pmc summary of the same kernels at this configuration."""
    stream = torch.cuda.current_stream()
    geom, net = fit.geom, fit.net
    N = geom.H * geom.W
    fwd, step_ref = flops_per_pixel(geom.F, net.bc, net.C, net.nl)
    # FLOPs are counted as EXECUTED: the fused step leaves the always-zero window-centre features out of its products
    # (lbdrn_train_step_features: 192 of 200 at the headline shape); the reference's count rides along
    Fe = ops.train_step_features(geom, net) if path != ops._lib.PATH_GENERIC else geom.F
    _, step = flops_per_pixel(Fe, net.bc, net.C, net.nl)
    ws = ops.ApplyWorkspace(geom, net, img_d.device)
    p = fit.best_params
    fast = codec.fast_evaluation()    # the arithmetic the fit's own evaluation passes ran in
    x16_on = fast and codec.exact16_evaluation()     # (only the opt-in leg: LBDRN_EVAL_X16=1)
    ops.eval_sse(geom, net, img_d, fit.msb, p, path, ws, fast=fast, x16=x16_on)
    t_eval = event_time_ms(lambda: ops.eval_sse(geom, net, img_d, fit.msb, p, path, ws, fast=fast, x16=x16_on), stream, 3)
    nsteps = (N + a.bs - 1) // a.bs
    peak = 157.3  # TFLOP/s, f32 MFMA == f32 vector peak (MI355X_MICROARCH.md)
    B = min(a.bs, N)
    out = {"bound": "mfma", "peak": peak, "unit": "TFLOP/s", "traffic": None}
    # the launch the timed region made: with four or more tiles in flight the fits of one shape step in PAIRS (one launch of
    # 2 x 128 workgroups per minibatch, one reduce launch for both: codec.fit_many); the probe times that launch, and the
    # launch of a fit alone beside it
    if per_launch is None:
        per_launch = tiles_per_launch(a, ops, path)

    def probe(count, alone=False):
        """per step, in ms, for `count` fits per launch: (whole step, training launch alone = mode 3, one more training
        launch = mode 2 - mode 0, one more reduce launch = mode 1 - mode 0).  alone: with the LBDRN_TRAIN_ALONE hint, as
        codec.fit_device steps a fit that has the device to itself (k_train_split where the shape has it)"""
        perms = [torch.randperm(N, device=img_d.device) for _ in range(count)]
        st = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for _ in range(count)]
        wss = [ops.TrainWorkspace(geom, net, a.bs, img_d.device).prepare(img_d, fit.msb, path) for _ in range(count)]
        if count == 1:
            run = lambda: ops.train_epoch(geom, net, img_d, fit.msb, perms[0], a.bs, *st[0], 0, 1e-6, None, path, wss[0], alone=alone)
        else:
            run = lambda: ops.train_epoch_group([geom] * count, net, [img_d] * count, [fit.msb] * count, perms, a.bs,
                                                [x[0] for x in st], [x[1] for x in st], [x[2] for x in st], 0, 1e-6,
                                                None, path, wss)
        run()
        t = {}
        try:
            for mode in (0, 3, 2, 1, 3, 0):     # plain, training launches alone, training doubled, reduce doubled, again
                ops.train_profile_mode(mode)
                t.setdefault(mode, []).append(event_time_ms(run, stream, 1))
        finally:
            ops.train_profile_mode(0)
        t_epoch = min(t[0])
        if net.bc > 64 and count == 1:   # the bc >= 128 step has two launches in front of its reduce launch: forward/backward alone
            ops.train_profile_mode(4)
            try:
                split_fb.append(min(event_time_ms(run, stream, 1) for _ in range(2)) / nsteps)
            finally:
                ops.train_profile_mode(0)
        return t_epoch / nsteps, min(t[3]) / nsteps, (t[2][0] - t_epoch) / nsteps, (t[1][0] - t_epoch) / nsteps

    split_fb = []
    t_step, t_own, t_train, t_reduce = probe(per_launch)
    fused = t_reduce > 0.5e-3   # a fused MFMA train kernel is in use (the generic path ignores the modes)
    wide = net.bc > 64
    name = "k_train_half + k_dw_wide" if wide else "k_train_stream"
    rows_wg = 32 if wide else 64
    nwg = (B + rows_wg - 1) // rows_wg
    # k_dw_wide's grid (csrc/train_wide.inc:dispatch_dw): (unit blocks x input blocks of every layer + the output layer) x slices of 1024 rows
    um, im0 = net.bc // 64, ((Fe + 15) // 16 + 3) // 4
    dw_wgs = (um * im0 + (um * um if net.nl > 1 else 0) + 1) * ((B + 1023) // 1024)
    key = config_key(a)
    prof, whole = committed_profile(key if N == 2048 * 2048 else None, per_launch)
    if fused:
        t_k = t_own
        out.update({"kernel": (f"{name} (the two training launches of a step: row gather + forward + loss + backward of one {B}-row "
                               f"minibatch on {nwg} workgroups of 32 rows -- units halved between two waves --, then its weight "
                               f"gradients as a batch-dimension GEMM on {dw_wgs} workgroups; every CU)" if wide else
                               f"{name} (row gather + forward + loss + backward + weight gradients of one {B}-row "
                               f"minibatch of {per_launch} fit(s): {per_launch} x {nwg} workgroups of 64 rows, one per CU, on "
                               f"{min(256, per_launch * nwg)} of the chip's 256 CUs)"),
                    "fits_per_launch": per_launch,
                    "kernel_us": round(t_own * 1e3, 2), "marginal_us": round(t_train * 1e3, 2),
                    "reduce_adam_us": round(t_reduce * 1e3, 2),
                    "unaccounted_us": round((t_step - t_train - t_reduce) * 1e3, 2),
                    "flop_per_launch": per_launch * step * B, "flop_per_launch_reference_arithmetic": per_launch * step_ref * B,
                    "features_multiplied": Fe, "cus_occupied": min(256, per_launch * nwg),
                    **({"forward_backward_us": round(split_fb[0] * 1e3, 2), "weight_gradient_us": round((t_own - split_fb[0]) * 1e3, 2)}
                       if split_fb else {}),
                    "timing": "HIP events on the launch stream.  kernel_us: one 512-launch epoch of the training launch the "
                              "timed region made (fits_per_launch fits side by side), back to back without the reduce/Adam "
                              "launches, every launch on its own rows (lbdrn_train_profile_mode 3) -- the kernel's own average "
                              "duration, launch boundary included; achieved = flop_per_launch / kernel_us.  "
                              "train_step_pair_us: the real epoch (train, reduce/Adam, train, ...) per step; marginal_us / "
                              "reduce_adam_us: what one more launch of either kernel costs in that sequence (epoch with the "
                              "launch doubled; the doubled training launch finds its rows warm: a lower bound, not the "
                              "kernel); unaccounted_us the rest of the step.  rocprof_kernel_us: the committed rocprofv3 "
                              "--kernel-trace average of the same launch sequence (profiles/, see rocprof_source); frac = "
                              "min(frac_live, frac_rocprof)"})
        if with_single and per_launch > 1:   # the launches of a fit alone (what a single tile runs), same method
            # Most of a lone fit's steps carry the LBDRN_TRAIN_ALONE hint: 2 x nwg workgroups of 32 rows on every CU
            # (k_train_split, csrc/train_split.inc; two gradient slabs per 64-row group) -- the same numbers bit for bit as
            # the half-chip launch, which the fit still uses beside its background evaluation pass (`half_chip_launch`)
            def rec(s1, o1, k1, r1, cus):
                a1 = step * B / (o1 * 1e-3) / 1e12
                return {"kernel_us": round(o1 * 1e3, 2), "marginal_us": round(k1 * 1e3, 2), "reduce_adam_us": round(r1 * 1e3, 2),
                        "unaccounted_us": round((s1 - k1 - r1) * 1e3, 2), "train_step_pair_us": round(s1 * 1e3, 2),
                        "flop_per_launch": step * B, "cus_occupied": cus, "achieved": round(a1, 3), "frac": round(a1 / peak, 4),
                        "frac_of_occupied_cus": round(a1 / (peak * cus / 256.0), 4)}
            half = rec(*probe(1), min(256, nwg))
            full = rec(*probe(1, alone=True), min(256, 2 * nwg))
            split_runs = full["kernel_us"] < 0.9 * half["kernel_us"]      # (shapes without k_train_split ignore the hint)
            skt = whole.get("configs", {}).get(key, {}).get("kernel_trace", {}) if (split_runs and key) else {}
            if skt.get("split_avg_us"):   # the committed rocprofv3 --kernel-trace average of the every-CU launch, as for the headline launch
                r_ach = step * B / (float(skt["split_avg_us"]) * 1e-6) / 1e12
                full.update({"rocprof_kernel_us": skt["split_avg_us"], "rocprof_kernel_min_us": skt.get("split_min_us"),
                             "rocprof_source": skt.get("source"), "frac_live": full["frac"], "frac_rocprof": round(r_ach / peak, 4),
                             "frac": round(min(full["achieved"], r_ach) / peak, 4)})
            out["single_fit_launch"] = dict(full if split_runs else half,
                                            kernel=(f"k_train_split: {2 * nwg} workgroups of 32 rows, units halved between two waves, every CU"
                                                    if split_runs else f"k_train_stream: {nwg} workgroups of 64 rows"),
                                            half_chip_launch=half if split_runs else None)
    else:  # shape without a fused train kernel: the generic step is many launches
        t_k = t_step
        out.update({"kernel": "generic train step (all launches of one minibatch)", "kernel_us": round(t_k * 1e3, 2)})
    ach = per_launch * step * B / (t_k * 1e-3) / 1e12
    # the same pass with layer 0's colour features on the f16 matrix pipe, operands exact (LBDRN_EVAL_X16: OPT-IN, never part of
    # `value`) -- what it would buy, measured
    t_eval_x16 = None
    if fast and path != ops._lib.PATH_GENERIC:
        ops.eval_sse(geom, net, img_d, fit.msb, p, path, ws, fast=True, x16=True)
        t_eval_x16 = event_time_ms(lambda: ops.eval_sse(geom, net, img_d, fit.msb, p, path, ws, fast=True, x16=True), stream, 3)
    out.update({"achieved": round(ach, 3), "frac": round(ach / peak, 4), "frac_live": round(ach / peak, 4),
                "train_step_pair_us": round(t_step * 1e3, 2),
                "train_step_pair_tflops": round(per_launch * step * B / (t_step * 1e-3) / 1e12, 3),
                "apply_pass_ms": round(t_eval, 3), "apply_tflops": round(fwd * N / (t_eval * 1e-3) / 1e12, 3),
                "apply_frac": round(fwd * N / (t_eval * 1e-3) / 1e12 / peak, 4),
                "apply_hbm_algorithmic_GBps": round(16.0 * N / (t_eval * 1e-3) / 1e9, 1),
                "apply_pass": "the per-epoch evaluation pass as the fit runs it: " +
                              ("LBDRN_EVAL_FAST (tolerance arithmetic)" + (" + LBDRN_EVAL_X16 (opt-in)" if x16_on else "") if fast else "canonical arithmetic"),
                "apply_pass_x16_ms_opt_in": round(t_eval_x16, 3) if t_eval_x16 is not None else None})
    if fused:
        # the kernel holds half of the chip (two fits' steps run side by side): the same rate against the peak of
        # the CUs it occupies
        out["frac_of_occupied_cus"] = round(ach / (peak * out["cus_occupied"] / 256.0), 4)
    if prof and fused:
        tr, ap = prof.get("train", {}), prof.get("apply_eval", {})
        pair = per_launch == 2 and (key == "bc64" or prof.get("_pair_counters"))
        out["traffic"] = tr.get("hbm_bytes_per_launch")
        out["traffic_algorithmic_bytes"] = B * 2 * net.C * (2 if pair else 1)   # one read of the uint16 source: 2 bytes a band
        out["traffic_note"] = ("counter values are per launch of a pair of fits (scripts/prof_pair.py)" if pair
                               else "counter values are per launch of ONE fit's minibatch (scripts/prof_fit.py runs one fit at a time)")
        out["traffic_kernel"] = tr.get("kernel")
        out["mfma_busy_frac"] = tr.get("mfma_busy_frac_whole_chip")
        out["mfma_busy_frac_occupied_simds"] = tr.get("mfma_busy_frac_occupied_simds")
        out["lds_bank_conflict_cycles_per_launch"] = tr.get("lds_bank_conflict_cycles_per_launch")
        out["apply_mfma_busy_frac"] = ap.get("mfma_busy_frac_whole_chip")
        out["counters_source"] = whole.get("source")
        kt = prof.get("kernel_trace", {})      # the committed rocprofv3 --kernel-trace --stats average of this launch
        if kt.get("train_avg_us"):
            r_us = float(kt["train_avg_us"]) + float(kt.get("dw_avg_us") or 0.0)   # (bc >= 128: both training launches)
            r_ach = per_launch * step * B / (r_us * 1e-6) / 1e12
            out.update({"rocprof_kernel_us": round(r_us, 2), "rocprof_kernel_min_us": kt.get("train_min_us"),
                        "rocprof_source": kt.get("source"), "frac_rocprof": round(r_ach / peak, 4),
                        "frac": round(min(ach, r_ach) / peak, 4)})
            if kt.get("timed_region_avg_us"):
                # ... and the same kernel's average in the TIMED configuration (two chains of such launches in flight take turns
                # on a chip either of them fills; the evaluation passes, row builds and permutations of the other fits share
                # it too): what `frac` -- measured on the launch sequence alone -- is not
                t_us = float(kt["timed_region_avg_us"]) + float(kt.get("timed_region_dw_avg_us") or 0.0)
                out.update({"rocprof_kernel_us_timed_region": round(t_us, 2), "rocprof_timed_region_source": kt.get("timed_region_source"),
                            "frac_timed_region": round(per_launch * step * B / (t_us * 1e-6) / 1e12 / peak, 4)})
        for extra in ("timeline", "stamps"):   # in-kernel evidence (s_memrealtime builds), committed as data
            if prof.get(extra):
                out[extra] = prof[extra]
    return out


def cpu_baseline(a):
    """The torch-CPU restatement of the reference loop (oracle/torch_port.py) on a crop of the same synthetic
    tile, bounded to ~10-30 s: `--cpu-epochs` epochs of the recipe (each = one shuffled training pass + one
    whole-image evaluation pass, identical work every epoch) on a `--cpu-sample`^2 crop, scaled linearly to the
    full `-e` epochs.  Form A (the headline): map-style dataset + DataLoader(shuffle, bs,
    num_workers=min(32, cores)) + per-step Adam + concatenating whole-image metric, i.e. the reference's cost
    structure; form B beside it: index_select minibatches + streaming MSE (math only).  GDAL / fpzip / file I/O
    excluded from both.  The CPU's best is what is reported: form B is timed at torch.set_num_threads in
    {8, 16, 32, 64, all} (a 64 x 200 GEMM on 8192 rows does not use 128 threads well), form A then runs at the thread
    count that served form B best; every measurement and its thread count are in the line."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    import torch_port as TP
    from lbdrn_hip.synth import synthetic_tile
    side = min(a.cpu_sample, a.height, a.width)
    img = synthetic_tile(0, a.bands, a.height, a.width)[:, :side, :side].copy()
    ocfg = O.FeatCfg(use_coordinates=a.coords_embedding, embedding=a.coords_embedding)
    ep = max(2, min(a.cpu_epochs, a.epochs)) if a.epochs > 1 else 1
    workers = min(32, os.cpu_count() or 1)
    all_threads = torch.get_num_threads()

    def run(faithful, threads):
        torch.set_num_threads(threads)
        torch.manual_seed(SEED)
        t0 = time.time()
        r = TP.fit(img, a.K, a.D, a.bc, a.nl, a.lr, a.bs, ep, cfg=ocfg, faithful=faithful,
                   num_workers=workers if faithful else 0)
        t_call = time.time() - t0
        t_loop = r["seconds"]                                # the epochs; the rest is the once-per-image feature build
        t_enc = (t_call - t_loop) + t_loop * (a.epochs / ep)
        t0 = time.time()
        TP.apply(r["msb"], r["params"], a.K, a.D, a.bc, a.nl, cfg=ocfg)
        t_dec = time.time() - t0
        return side * side / (t_enc + t_dec) / 1e6, t_enc, t_dec, t_call

    full = None
    try:
        sweep = {}
        for th in sorted({t for t in (8, 16, 32, 64, all_threads) if t <= all_threads}):
            sweep[th] = run(False, th)
        best_th = max(sweep, key=lambda th: sweep[th][0])
        form_a = run(True, best_th)
        if a.cpu_full_epochs > 0 and side < min(a.height, a.width):
            # form B once more on the WHOLE tile, nothing scaled but (optionally) the epoch count: ~30 s at the headline shape
            torch.set_num_threads(best_th)
            torch.manual_seed(SEED)
            whole = synthetic_tile(0, a.bands, a.height, a.width)
            epf = max(2, min(a.cpu_full_epochs, a.epochs)) if a.epochs > 1 else 1
            t0 = time.time()
            r = TP.fit(whole, a.K, a.D, a.bc, a.nl, a.lr, a.bs, epf, cfg=ocfg, faithful=False, num_workers=0)
            t_call = time.time() - t0
            t_enc = (t_call - r["seconds"]) + r["seconds"] * (a.epochs / epf)
            t0 = time.time()
            TP.apply(r["msb"], r["params"], a.K, a.D, a.bc, a.nl, cfg=ocfg)
            t_dec = time.time() - t0
            full = {"mpixels_per_s": round(a.height * a.width / (t_enc + t_dec) / 1e6, 6), "threads": best_th, "epochs_measured": epf,
                    "measured_s": round(t_call + t_dec, 1), "encode_s": round(t_enc, 1), "decode_s": round(t_dec, 1),
                    "sample": f"the whole {a.bands} x {a.height} x {a.width} tile, {epf} of {a.epochs} epochs measured"}
            del whole, r
    finally:
        torch.set_num_threads(all_threads)
    fb = sweep[best_th]
    return {"value": round(form_a[0], 6), "unit": "Mpixels/s", "cores": best_th,
            "host_cpus": os.cpu_count(), "kind": "port", "loader_workers": workers,
            "value_note": "form A on the crop, scaled linearly to the tile: an UPPER bound of what this CPU does on the whole tile "
                          "(the reference's concatenating metric is quadratic in the number of minibatches: the crop understates "
                          "its cost); form B was also run on the whole tile (vectorised_form_B_full_tile)",
            "vectorised_form_B": round(fb[0], 6),
            "vectorised_form_B_full_tile": full,
            "form_B_by_threads": {str(th): round(v[0], 6) for th, v in sweep.items()},
            "sample": f"{side}x{side}x{a.bands} crop of tile 0 ({(side * side + a.bs - 1) // a.bs} minibatches per pass), "
                      f"{ep} of {a.epochs} epochs measured (bs={a.bs}; every epoch = one training pass + one evaluation "
                      f"pass) and scaled to {a.epochs}; torch threads swept over {sorted(sweep)} on form B (index_select "
                      f"batches + streaming MSE), best at {best_th}: measured {fb[3]:.1f}s -> encode {fb[1]:.1f}s decode "
                      f"{fb[2]:.1f}s; form A (the value) = DataLoader(num_workers={workers}) + per-step Adam + "
                      f"concatenating eval metric at {best_th} threads: measured {form_a[3]:.1f}s -> encode {form_a[1]:.1f}s "
                      f"decode {form_a[2]:.1f}s"}


def tile_flops(a, ops, fit, path):
    """Algorithmic FLOPs of one tile as executed: `epochs` training passes (the fused step's feature count), as many
    evaluation passes (none for a one-epoch fit) and the decode pass over all F features."""
    Fe = ops.train_step_features(fit.geom, fit.net) if path != ops._lib.PATH_GENERIC else fit.geom.F
    fwd_f, _ = flops_per_pixel(fit.geom.F, fit.net.bc, fit.net.C, fit.net.nl)
    _, step_f = flops_per_pixel(Fe, fit.net.bc, fit.net.C, fit.net.nl)
    evals = a.epochs if a.epochs > 1 else 0
    return a.height * a.width * (a.epochs * step_f + (evals + 1) * fwd_f)


def lone_tile(codec, ops, tile, a, path, laps_n=3):
    """One tile alone on the device (nothing else in flight): encode fit | truncation + decode, median of `laps_n`."""
    one = argparse.Namespace(**vars(a))
    one.in_flight = 1
    run_images(codec, ops, [tile], one, path)      # untimed: this stream's allocator pool has not held a workspace yet
    torch.cuda.synchronize()
    laps = []
    for _ in range(laps_n):
        ts = time.perf_counter()
        lone = codec.fit_many([tile], a.K, a.D, a.bc, a.nl, a.lr, a.bs, a.epochs, cfg=feat_cfg(a), path=path,
                              seed=SEED, in_flight=1)[0]
        torch.cuda.synchronize()
        tm = time.perf_counter()
        rec = codec.apply_device(lone.geom, lone.net, lone.msb, codec.truncate_device(lone.best_params, 16), path=path)
        torch.cuda.synchronize()
        te = time.perf_counter()
        laps.append(((te - ts) * 1e3, (tm - ts) * 1e3, (te - tm) * 1e3))
    laps.sort()
    ms, enc, dec = laps[len(laps) // 2]
    return {"ms": ms, "encode_ms": enc, "decode_ms": dec, "fit": lone, "rec": rec}


def side_leg(codec, ops, tiles, a, path, label, with_single=False, env=None, **over):
    """A short leg of another BASELINE.json configuration inside the same call: warm-up tiles twice per in-flight stream,
    ONE timed region of `steps` tiles with `in_flight` progressing together, one tile alone (median of 3), the training
    launch's own duration (roofline_probe without the single-fit detour).  Same tiles, same method, same clock as the
    headline; one repeat instead of three."""
    b = argparse.Namespace(**vars(a))
    for k, v in over.items():
        setattr(b, k, v)
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        return _side_leg(codec, ops, tiles, b, path, label, with_single)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _side_leg(codec, ops, tiles, b, path, label, with_single):
    t_leg = time.perf_counter()
    take = [tiles[k % len(tiles)] for k in range(b.warmup + b.steps)]     # (inputs are read-only: a tile may serve twice)
    for _ in range(2):
        run_images(codec, ops, take[:b.warmup], b, path)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = run_images(codec, ops, take[b.warmup:], b, path)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fit, rec = done[-1]
    img_d = take[-1]
    diff = (img_d.view(torch.int16).to(torch.int32) & 0xFFFF).float() - (rec.to(torch.int32) & 0xFFFF).float()
    mse = float((diff * diff).mean().item())
    single = lone_tile(codec, ops, img_d, b, path)
    same = bool(torch.equal(single["fit"].best_params.view(torch.int32), fit.best_params.view(torch.int32))
                and torch.equal(single["rec"], rec))
    px = b.height * b.width
    tile_flop = tile_flops(b, ops, fit, path)
    roof = roofline_probe(codec, ops, fit, img_d, b, path, with_single=with_single)
    keep = ("kernel", "fits_per_launch", "kernel_us", "forward_backward_us", "weight_gradient_us", "marginal_us", "reduce_adam_us",
            "unaccounted_us", "train_step_pair_us", "features_multiplied",
            "flop_per_launch", "achieved", "frac", "frac_live", "frac_rocprof", "rocprof_kernel_us", "rocprof_source",
            "rocprof_kernel_us_timed_region", "frac_timed_region", "rocprof_timed_region_source", "single_fit_launch", "apply_pass_x16_ms_opt_in",
            "frac_of_occupied_cus", "apply_pass_ms", "apply_frac", "traffic", "mfma_busy_frac", "mfma_busy_frac_occupied_simds")
    out = {"workload": label, "steps": b.steps, "warmup": b.warmup, "tiles_in_flight": min(b.in_flight, b.steps),
           "tiles_per_launch": tiles_per_launch(b, ops, path), "repeats": 1,
           "ms_per_step": round(elapsed / b.steps * 1e3, 3), "mpixels_per_s": round(px * b.steps / elapsed / 1e6, 4),
           "single_tile_ms": round(single["ms"], 3), "single_tile_encode_ms": round(single["encode_ms"], 3),
           "single_tile_decode_ms": round(single["decode_ms"], 3),
           "end_to_end_frac": round(tile_flop * b.steps / elapsed / 1e12 / 157.3, 4),
           "single_tile_end_to_end_frac": round(tile_flop / (single["ms"] * 1e-3) / 1e12 / 157.3, 4),
           "timed_equals_lone": same, "recon_mse_last_tile": round(mse, 4),
           "roofline": {k: roof[k] for k in keep if k in roof},
           "accounting": accounting(b, roof, tiles_per_launch(b, ops, path), elapsed / b.steps * 1e3, single["decode_ms"])}
    del single, done, fit, rec
    torch.cuda.empty_cache()      # the next leg's workspaces have other sizes
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    return out


def accounting(a, roof, fits_per_launch, ms_per_tile, decode_ms):
    """Why a tile takes what it takes, as an equation over measured kernel durations.  With several
    fits in flight every launch below wants every CU, so they take turns and their durations ADD:
        tile ~= (steps / fits per launch) x training launch  +  evaluation passes  +  decode pass  +  rest
    `training launch` = the launch's own duration (roofline.kernel_us: the launch sequence alone), `rest` = what is left of
    the measured tile: the row build, the permutations, reduce / Adam launches that found no room beside another chain's
    training launch, ramps.  An identity at the measured operating point, not a law: timing-only builds
    (profiles/r06_bounds_in_flight.txt) show the tile sitting where three bounds coincide -- the matrix pipe (this launch's own
    duration), the memory system under two chains of launches (gradient slabs, gathered rows) and the latency of a chain's step --,
    so a faster training launch ALONE moves the tile by far less than its term here (-18 % of the launch: -2 % of the tile)."""
    px = a.height * a.width
    steps = a.epochs * ((px + a.bs - 1) // a.bs)
    launches = steps / max(fits_per_launch, 1)
    evals = a.epochs if a.epochs > 1 else 0
    train_ms = launches * roof.get("kernel_us", 0.0) * 1e-3
    eval_ms = evals * roof.get("apply_pass_ms", 0.0)
    known = train_ms + eval_ms + decode_ms
    return {"training_launches_per_tile": launches, "training_ms": round(train_ms, 2), "evaluation_passes_ms": round(eval_ms, 2),
            "decode_ms": round(decode_ms, 2), "rest_ms": round(ms_per_tile - known, 2), "measured_ms_per_tile": round(ms_per_tile, 2),
            "training_share": round(train_ms / ms_per_tile, 3),
            "equation": "measured_ms_per_tile = training_launches_per_tile x roofline.kernel_us + epochs x roofline.apply_pass_ms + decode_ms + rest_ms",
            "note": "an identity at this operating point, where the matrix pipe, the memory system (slabs, rows) and a chain's step latency "
                    "bound the tile together: profiles/r06_bounds_in_flight.txt, DESIGN.md 4.0"}


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N ranks, one per GPU, BEFORE this process makes
    any GPU call (it never does), relay rank 0's JSON line, fail if any rank fails.  Same environment contract as
    torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import socket
    import subprocess
    backend = os.environ.get("LBDRN_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()          # does not initialise the GPU
    if backend == "nccl" and ndev < a.gpus and os.environ.get("LBDRN_BENCH_DRYRUN") != "1":
        print(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryDirectory(prefix="lbdrn_bench_") as logdir:
        logs = []
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            # every rank's stdout and stderr are kept: when a rank fails, what IT said is what gets printed
            out_f = open(os.path.join(logdir, f"rank{r}.out"), "w+")
            err_f = open(os.path.join(logdir, f"rank{r}.err"), "w+")
            logs.append((out_f, err_f))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out_f, stderr=err_f))
        # a rank that dies leaves the others waiting in a rendezvous or a barrier: stop them (by PID) right away
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                break
            time.sleep(0.1)
        codes = [p.wait() for p in procs]

        def text(f):
            f.flush()
            f.seek(0)
            return f.read()
        for ln in text(logs[0][0]).splitlines():      # the JSON line only (gloo announces its connections on stdout)
            if ln.startswith("{"):
                print(ln)
            elif ln.strip():
                print(ln, file=sys.stderr)
        sys.stdout.flush()
        if any(codes):
            print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
            first_bad = [r for r, c in enumerate(codes) if c not in (0, -15)] or [r for r, c in enumerate(codes) if c]
            for r in first_bad[:2]:
                tail = "\n".join((text(logs[r][1]) + text(logs[r][0]) if r else text(logs[r][1])).splitlines()[-25:])
                print(f"---- rank {r} (exit code {codes[r]}), last lines:\n{tail}", file=sys.stderr)
        else:
            sys.stderr.write(text(logs[0][1]))
        for o, e in logs:
            o.close()
            e.close()
    return 1 if any(codes) else 0


def dry_run(a, rank, world):
    """LBDRN_BENCH_DRYRUN=1: the launcher and the exchange only (gloo, no GPU call) -- what the CPU test of
    `--gpus 2` runs; prints the shape of the bench line with no measurement in it."""
    from lbdrn_hip import shard
    if os.environ.get("LBDRN_BENCH_DRYRUN_FAIL") == str(rank):   # rehearsal of a failing rank (tests)
        print(f"rank {rank}: simulated failure before the rendezvous", file=sys.stderr)
        return 7
    if world > 1:
        dist.init_process_group("gloo")
    mine = shard.assign((a.warmup + a.steps) * world, rank, world)
    records = shard.gather_records([[float(i), 0.0, 0.0] for i in mine[a.warmup:]], 3)
    elapsed = shard.max_over_ranks(1.0 + rank)
    per_rank = shard.gather_records([[float(rank), 1000.0 * (1.0 + rank)]], 2)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "records": records,
                          "max_over_ranks": elapsed, "rank_elapsed_ms": [r[1] for r in sorted(per_rank)],
                          "ranks_seen": dist.get_world_size() if world > 1 else 1}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    a = parse()
    launched = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not launched:
        return launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        return 2
    if os.environ.get("LBDRN_BENCH_DRYRUN") == "1":
        return dry_run(a, rank, world)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (liblbdrn_hip has no CPU path)", file=sys.stderr)
        return 2
    # one process per GPU.  LBDRN_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks
    # (ranks then share devices round-robin and the three scalars per rank travel over gloo instead of RCCL)
    backend = os.environ.get("LBDRN_BENCH_BACKEND", "nccl")
    if backend == "nccl" and local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU of its own ({torch.cuda.device_count()} visible)", file=sys.stderr)
        return 2
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    xdev = dev if backend == "nccl" else None   # where the exchanged scalars live
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from lbdrn_hip import codec, ops, shard
    from lbdrn_hip.synth import synthetic_tile
    path = {"auto": ops._lib.PATH_AUTO, "generic": ops._lib.PATH_GENERIC, "mfma": ops._lib.PATH_MFMA}[a.path]

    # the job is (warmup + steps) * world images, dealt round-robin; this rank's tiles are
    # resident in HBM before the clock starts
    total = a.warmup + a.steps
    mine = shard.assign(total * world, rank, world)
    tiles = [ops.to_device_u16(synthetic_tile(i, a.bands, a.height, a.width), dev) for i in mine]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # W warm-up steps on every in-flight stream (each stream has its own allocator pool): the W warm-up tiles go through
    # twice per stream -- on a fresh box the first process still pays one-time costs in the second fit of a stream
    # (measured on configs[4]: 154 instead of 96 ms per tile with a single round)
    for _ in range(2 if a.warmup else 0):
        run_images(codec, ops, tiles[:a.warmup] * max(1, min(a.in_flight, a.steps)), a, path)
    # the timed region -- exactly K tiles per GPU between two barrier + synchronize pairs, maximum over the ranks -- runs
    # `--repeats` times (default 3); the line carries the MEDIAN repeat and the spread
    rep_laps, own = [], []
    for _ in range(a.repeats):
        barrier()
        t0 = time.perf_counter()
        done = run_images(codec, ops, tiles[a.warmup:], a, path)
        barrier()
        own.append(time.perf_counter() - t0)
        rep_laps.append(shard.max_over_ranks(own[-1], xdev))
    order = sorted(range(a.repeats), key=lambda k: rep_laps[k])
    mid = order[len(order) // 2]
    elapsed = rep_laps[mid]
    rank_elapsed = shard.gather_records([[float(rank), own[mid] * 1e3]], 2, xdev)   # every rank's own clock, that repeat

    # per-image metric records (after the clock): [image index, reconstruction MSE, best evaluation MSE] for every
    # timed tile -- the only data the ranks exchange
    recs = []
    for idx, img_d, (fit, rec) in zip(mine[a.warmup:], tiles[a.warmup:], done):
        diff = (img_d.view(torch.int16).to(torch.int32) & 0xFFFF).float() - (rec.to(torch.int32) & 0xFFFF).float()
        recs.append([float(idx), float((diff * diff).mean().item()), float(fit.mse_log[:, 0].min().item())])
    records = shard.gather_records(recs, 3, xdev)
    fit, rec = done[-1]
    img_d, mse = tiles[-1], recs[-1][1]

    # one tile alone on the GPU (BASELINE.json configs[1] is "a single tile"): rank 0, after the clock
    single = None
    if rank == 0:
        single = lone_tile(codec, ops, tiles[-1], a, path)
        # the launch sequence the timed region runs (pairs of fits per launch, four host threads) against the same fit
        # alone: bit-identical weights and raster, or the line says so
        lone, lone_rec = single["fit"], single["rec"]
        timed_equals_lone = bool(torch.equal(lone.best_params.view(torch.int32), fit.best_params.view(torch.int32))
                                 and torch.equal(lone_rec, rec))

    if rank == 0:
        px = a.height * a.width
        value = px * a.steps * world / elapsed / 1e6
        out = {
            "metric": "Mpixels/s encode+decode (and bpp/PSNR parity) on D2/K5/bc64/nl2",
            "value": round(value, 4), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "repeats": a.repeats, "ms_per_step_all_repeats": [round(t / a.steps * 1e3, 3) for t in rep_laps],
            "rank_elapsed_ms": [round(r[1], 3) for r in sorted(rank_elapsed)],
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"synthetic {a.bands}-band {a.height}x{a.width} uint16 tile per step, "
                                   f"K={a.K} D={a.D} bc={a.bc} nl={a.nl} bs={a.bs} e={a.epochs}"
                                   f"{' USE_COORDINATES+EMBEDDING' if a.coords_embedding else ''} "
                                   f"(BASELINE.json configs[1] by default); encode fit + 16-bit weight truncation + decode",
                       "tiles_per_gpu": a.steps, "tiles_in_flight_per_gpu": min(a.in_flight, a.steps),
                       "tiles_per_launch": tiles_per_launch(a, ops, path),
                       "warmup_note": "the warm-up tiles run twice on each in-flight stream",
                       "parallelism": f"image-sharded x{world}", "path": a.path},
            "single_tile_ms": round(single["ms"], 3),
            "single_tile_encode_ms": round(single["encode_ms"], 3), "single_tile_decode_ms": round(single["decode_ms"], 3),
            "single_tile_mpixels_per_s": round(px / single["ms"] / 1e3, 4),
            "single_tile_note": "one tile alone on one GPU (nothing else in flight), encode fit | truncation + decode, median of 3; "
                                "`value` is the throughput with tiles_in_flight_per_gpu independent tiles progressing together",
            "timed_equals_lone": timed_equals_lone,
            "timed_equals_lone_note": "the last timed tile's fit (as the timed region ran it: fits stepping in pairs per launch, "
                                      "several in flight) against the same tile fitted alone afterwards: best weights and decoded "
                                      "raster compared bit for bit",
            "recon_mse_last_tile": round(mse, 4),
            "recon_psnr_last_tile": round(10 * np.log10(10000 ** 2 / max(mse, 1e-12)), 3),
            "records": records,
        }
        out["roofline"] = roofline_probe(codec, ops, fit, img_d, a, path)
        # whole job against the matrix peak: algorithmic FLOPs of every timed tile (10 train + 10 evaluation passes + decode)
        tile_flop = tile_flops(a, ops, fit, path)
        out["roofline"]["end_to_end_tflops"] = round(tile_flop * a.steps * world / elapsed / 1e12, 3)
        out["roofline"]["end_to_end_frac"] = round(tile_flop * a.steps / elapsed / 1e12 / 157.3, 4)
        out["roofline"]["single_tile_end_to_end_frac"] = round(tile_flop / (single["ms"] * 1e-3) / 1e12 / 157.3, 4)
        out["roofline"].update(hbm_fields(a, elapsed / a.steps, config_key(a), out["config"]["tiles_per_launch"]))
        out["accounting"] = accounting(a, out["roofline"], out["config"]["tiles_per_launch"], elapsed / a.steps * 1e3, single["decode_ms"])
        del single, lone, lone_rec
        if world == 1 and not a.no_other_configs and config_key(a) == "bc64" and a.height == 2048 and a.width == 2048 \
                and a.path == "auto":
            # BASELINE.json configs[2] and configs[4], and the reference's own majority shape -- four bands (9 of the 13 images of
            # run.sh:14-28), the first four bands of the resident tiles --, a short leg each in the same call (the headline above is
            # unchanged)
            out["other_configs"] = {
                "bands4": side_leg(codec, ops, [t[:4] for t in tiles], a, path, with_single=True, bands=4, in_flight=4, steps=8, warmup=4,
                                   label="the reference's 4-band shape (GF-2 / GF6-PMS: run.sh:14-28), K=5 D=2 bc=64 nl=2, F = 100 "
                                         "(k_train_stream<24,2,3,6> in flight, k_train_split<24,6> alone)"),
                "bc256": side_leg(codec, ops, tiles, a, path, bc=256, in_flight=codec.default_in_flight(a.bands, a.height, a.width, a.K, a.D, 256, a.nl),
                                  steps=6, warmup=3,   # (codec.default_in_flight: 3 chains at bc >= 128 -- 2: 352, 3: 343, 4: 386, 6: 345 ms per tile)
                                  label="BASELINE.json configs[2]: the same tile, bc = 256 (k_train_half + k_dw_wide / k_apply_wide)"),
                "embed": side_leg(codec, ops, tiles, a, path, coords_embedding=True, in_flight=4, steps=8, warmup=4,
                                  label="BASELINE.json configs[4]: USE_COORDINATES + EMBEDDING (F = 250)"),
                # NOT the headline's arithmetic: the same tiles with the OPT-IN evaluation pass (LBDRN_EVAL_X16=1: layer 0's colour
                # features on the f16 matrix pipe, integer differences x W_0 in three fp16 pieces, every product exact, float32
                # sums; the training step and the decode pass as in the headline).  What DESIGN.md 10 prices, measured.
                "x16_opt_in": side_leg(codec, ops, tiles, a, path, in_flight=4, steps=8, warmup=4, env={"LBDRN_EVAL_X16": "1"},
                                       label="configs[1] with LBDRN_EVAL_X16=1 (opt-in; off in the headline): the per-epoch evaluation "
                                             "passes take layer 0 through the f16 matrix pipe with exact operands"),
            }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
