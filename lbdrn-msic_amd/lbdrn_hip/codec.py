"""Per-image fit (encode side) and apply (decode side) loops on the GPU.

fit_image() stands where the reference's encode.train() stands (ref encode.py:67-157) and
apply_image() where decode.test()'s numeric core stands (ref decode.py:73-134); file handling,
payload coding and logging stay in encode.py / decode.py.
"""
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops
from .features import FeatCfg
from .model import LBDRNModel
from .sampler import DevicePermutationStream, GPU_RANDPERM_MAX, draw_pass_seeds, epoch_plan


def lr_schedule(lr, epochs):
    """Learning rate in force during each epoch: StepLR(step_size=max(1,int(epochs/3)), gamma=0.1)
    stepped at every epoch end (ref encode.py:85,98); chained float64 products like torch's."""
    step = max(1, int(epochs / 3))
    out, cur = [], float(lr)
    for e in range(1, epochs + 1):
        out.append(cur)
        if e % step == 0:
            cur *= 0.1
    return out


def skip_fit_rng(n_feature, base_channel, channels, num_layers, epochs, val_duration=1):
    """Consume the global torch CPU generator exactly as one fit_device() / reference train() call does
    (model construction, then two draws per DataLoader iterator), without fitting.  The reference fits
    the split_ratio tiles of an image one after another on ONE generator (ref encode.py:200-205, 231-262),
    so tile t's initial weights and minibatch orders depend on how much tiles 0..t-1 drew; a rank that
    fits only some of the tiles calls this for the ones it leaves to other ranks and so reproduces the
    serial run bit for bit."""
    draw_fit(n_feature, base_channel, channels, num_layers, epochs, val_duration)


class FitDraws:
    """Everything one fit takes from the global torch generator, drawn ahead of time: the initial parameters
    (state_dict order) and the sampler seeds of its training passes."""

    def __init__(self, params, train_seeds):
        self.params, self.train_seeds = params, train_seeds


def draw_fit(n_feature, base_channel, channels, num_layers, epochs, val_duration=1):
    """Make the draws of one fit now, on the calling thread (same consumption as skip_fit_rng /
    fit_device): the fits of the tiles of one image share ONE generator in the reference
    (ref encode.py:231-262), so they can only progress together (fit_many) if each one's draws are taken
    in tile order beforehand."""
    model = LBDRNModel(dim_in=n_feature, dim_hidden=base_channel, dim_out=channels, num_layers=num_layers)
    return FitDraws(model.flat_parameters(), draw_pass_seeds(epoch_plan(epochs, val_duration)))


class FitResult:
    def __init__(self):
        self.params = None          # best parameters, float32 numpy, state_dict order
        self.best_epoch = -1
        self.best_mse = None
        self.epoch_mse = []         # (epoch, mse, improved)
        self.losses = None          # per-step minibatch losses [epochs][steps] (device tensor)
        self.msb = None             # MSB plane, numpy uint8/uint16 [C,H,W] (None with host_msb=False)
        self.msb_device = None      # the same plane in HBM (uint16 bits in int16 storage)
        self.msb_max = None
        self.n_feature = None
        self.channels = None
        self.n_subpixels = None
        self.seconds = {}


class DeviceFit:
    """Result of fit_device(): everything still in HBM; host() syncs once."""

    def __init__(self):
        self.best_params = None   # device float32 [NP]
        self.msb = None           # device int16-storage uint16 [C,H,W]
        self.msb_max = None
        self.geom = None
        self.net = None
        self.mse_log = None       # device [epochs,2]: mse, improved flag
        self.evaluated = []
        self.losses = None
        self.epochs = 0


_RNG_LOCK = threading.Lock()   # the global torch CPU generator is one per process
_POOL_LOCK = threading.Lock()
_FIT_STREAMS = {}               # device -> streams the fits in flight run on


def _fit_stream_pool(dev, n):
    """The process-wide streams fits run on (kept for life: torch's allocator pools memory per stream), at least n."""
    with _POOL_LOCK:
        pool_ = _FIT_STREAMS.setdefault(dev, [])
        while len(pool_) < n:
            pool_.append(torch.cuda.Stream(device=dev))
        return pool_


def fast_evaluation():
    """The per-epoch evaluation passes of a fit rank its epochs (encode.py:104-117): an encode-time float under the
    1e-5 tolerance contract like the training loss, so they run in the training step's arithmetic (LBDRN_EVAL_FAST:
    hardware sin / exp behind a compensated reduction, 4.5e-7 per activation; the sum within 1e-6 relative of the
    canonical one, 20 % less time per pass).  LBDRN_EVAL_CANONICAL=1 keeps them on the decode kernels' canonical
    arithmetic (bit for bit the oracle's sum).  Decoding is always canonical."""
    return os.environ.get("LBDRN_EVAL_CANONICAL") != "1"


def exact16_evaluation():
    """LBDRN_EVAL_X16=1 (opt-in, off by default): the fast evaluation passes take layer 0's colour features through the f16
    matrix pipe with exact operands (lbdrn_hip.h: LBDRN_EVAL_X16) where the shape and the image qualify."""
    return os.environ.get("LBDRN_EVAL_X16") == "1"


def alone_streams_fit_queues():
    """A fit alone in its process uses three streams at a time: the caller's, one fit stream for the background passes,
    the permutation side stream."""
    return 3 <= int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))


def overlap_evaluation(dev, alone):
    """Whether a fit's evaluation passes run in the background of its next epoch (fit_device).  A stated policy, not a
    measurement: LBDRN_OVERLAP_EVAL=0 / 1 decides outright (run.sh and sweep.py export 0 when they put several
    processes on one GPU: `alone` only knows about this process); otherwise the passes go to the background when the
    fit is alone in its process AND the three streams it then works on -- the caller's, one fit stream, the
    permutation side stream -- can each have a hardware queue (GPU_MAX_HW_QUEUES, the runtime's default is 4): a
    background pass that shares a queue with the training chain costs 3x the fit instead of saving 5 % of it
    (scripts/ab_streamidx.py).  Results are bit-identical either way."""
    env = os.environ.get("LBDRN_OVERLAP_EVAL")
    if env in ("0", "1"):
        return env == "1"
    if not alone or device_shared():
        return False
    return alone_streams_fit_queues()


def device_shared():
    """LBDRN_DEVICE_SHARED=1: other PROCESSES fit on this GPU (run.sh / sweep.py with PER_GPU > 1 export it).  A process
    cannot see its neighbour's fits, so the caller says so: its fits then never claim the whole device -- no LBDRN_TRAIN_ALONE
    hint (the every-CU training launch would take turns with the neighbour's), evaluation passes in the chain."""
    return os.environ.get("LBDRN_DEVICE_SHARED", "0") not in ("", "0")


def fit_bytes(C, H, W, K, D, base_channel, num_layers, batch_size, epochs, cfg=None):
    """Device bytes one fit of this shape holds while it runs (fit_device): the training workspace -- dominated by the
    [N][F + C] row matrix, 832 B a pixel at the headline shape --, the permutations of all epochs and their scratch, the
    image, its MSB plane and the decoded raster, the evaluation workspace, snapshots.  What fit_many divides the free
    memory by."""
    cfg = cfg or FeatCfg.from_constants()
    N = H * W
    g = ops._lib.Geom(C, H, W, K, D, 1, int(cfg.use_colors), int(cfg.relative), int(cfg.P), 0, None, None)
    net = ops.make_net(cfg.feature_dim(C, D), base_channel, C, num_layers, cfg.act)
    L = ops.lib()
    import ctypes
    train = int(L.lbdrn_train_workspace(ctypes.byref(g), ctypes.byref(net), batch_size))
    apply_ws = int(L.lbdrn_apply_workspace(ctypes.byref(g), ctypes.byref(net)))
    perms = epochs * N * 8 + int(L.lbdrn_randperm_workspace(N, min(32, max(epochs - 1, 1))))
    planes = 3 * C * N * 2                       # image, MSB plane, reconstruction
    return train + apply_ws + perms + planes + (epochs + 4) * int(ops.param_count(net)) * 4 + (64 << 20)


def default_in_flight(C, H, W, K, D, base_channel, num_layers, cfg=None, path=None):
    """How many fits of this shape progress together on one GPU when the caller does not say (fit_many, the CLIs): as many
    chains as can each have a launch resident, one more would only queue.
    * the fused bc = 64 step takes pairs of fits per launch (2 x 128 workgroups of 147 KB of LDS = every CU, one workgroup
      each): two such chains alternate -- one trains while the other reduces / evaluates -- and saturate the chip: 4 fits;
    * the bc >= 128 step is three launches that all want every CU: k_train_half (72 KB of LDS, 194 registers: two
      workgroups per CU), k_dw_wide (66 KB, 284 registers: one wave per SIMD), k_reduce_adam / the evaluation pass.  Three
      chains can each have one of them resident; a fourth chain's forward/backward workgroups queue behind the resident pair
      and which chains meet decides the time (343 or 386 ms per tile: DESIGN 4.5): 3 fits;
    * the generic path is many small launches per step: 4."""
    cfg = cfg or FeatCfg.from_constants()
    if path == ops._lib.PATH_GENERIC:
        return 4
    if base_channel >= 128:
        return 3
    return 4


def memory_limited_in_flight(images, wanted, K, D, base_channel, num_layers, batch_size, epochs, cfg=None):
    """How many of `images` may progress together: `wanted`, or fewer where the device's free memory (plus what this
    process's allocator already holds in reserve) does not hold that many of the largest fit at 85 %.  One 8 x 2048^2 tile is
    3.9 GB, a 6000 x 6000 x 8 scene (the reference's GF6-WFI size) 34 GB; a fit that does not fit at all raises
    with the numbers (split the image: -sr)."""
    if not images:
        return wanted
    dev = images[0].device
    need = max(fit_bytes(*tuple(t.shape), K, D, base_channel, num_layers, batch_size, epochs, cfg) for t in images)
    free, total = torch.cuda.mem_get_info(dev)
    free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)     # cached blocks are this process's to reuse
    if device_shared():
        free //= 2                                                                  # (the neighbour process sizes itself the same way)
    fits = int(0.85 * free // need)
    if fits < 1:
        raise ops._lib.LbdrnError(f"one fit of this shape holds {need / 2**30:.1f} GiB on the device, {free / 2**30:.1f} GiB are free "
                                  f"(of {total / 2**30:.0f}): split the image (-sr)")
    return max(1, min(wanted, fits))


# What the two launches that share the chip during a lone fit's background evaluation pass achieve, as fractions of the
# f32-MFMA peak (157.3 TFLOP/s) -- measured, DESIGN.md section 4: the evaluation pass on the whole chip (it runs on half of
# it in the background: twice as long), and k_train_stream's single-fit launch on its 128 CUs, plus the step's two kernel
# boundaries and its reduce / Adam launch.  These are the FIRST GUESS only (round 6): every lone fit times its first background
# pass and the steps beside it with events it never waits for, and the next fit of that shape in the process draws the
# line where the measurement puts it (_calibrated_head) -- constants that go stale with a kernel change now cost the first
# fit of a process a few per cent, not every fit (tests/test_gpu_train_paths.py holds the guess to the measurement).
EVAL_PASS_FRAC_OF_PEAK, HALF_CHIP_STEP_FRAC_OF_PEAK, HALF_CHIP_STEP_OVERHEAD_S, PEAK_FLOPS = 0.65, 0.26, 5.0e-6, 157.3e12
_HEAD_LOCK = threading.Lock()
_HEAD_MEASURED = {}     # shape key -> steps that last as long as the background pass beside them, as measured
_HEAD_PENDING = {}      # shape key -> ((pass start, pass end, head start, head end) events, head steps) of a fit that has been enqueued


def _head_key(dev, net, n_pixels, batch_size):
    return (dev.index, int(net.F), int(net.bc), int(net.C), int(net.nl), int(net.act), int(n_pixels), int(batch_size))


def _calibrated_head(key, guess, steps_per_epoch):
    """The head count of `key`: the model's guess until a fit of that shape has run, then what its events say -- the head
    scaled by pass time / head time, kept within [guess / 2, 2 guess].  Never waits: events that have not completed yet are
    looked at by a later fit."""
    with _HEAD_LOCK:
        pend = _HEAD_PENDING.get(key)
        if pend is not None and pend[0][1].query() and pend[0][3].query():
            (p0, p1, h0, h1), head = pend
            del _HEAD_PENDING[key]
            t_pass, t_head = p0.elapsed_time(p1), h0.elapsed_time(h1)
            if t_pass > 0 and t_head > 0 and head > 0:
                _HEAD_MEASURED[key] = int(round(min(max(head * t_pass / t_head, 0.5 * guess), 2.0 * guess)))
        return max(0, min(steps_per_epoch, _HEAD_MEASURED.get(key, guess)))


def head_calibration():
    """{shape key: measured head steps} of this process so far (diagnostics, tests)."""
    with _HEAD_LOCK:
        return dict(_HEAD_MEASURED)


def background_steps(steps_per_epoch, net=None, n_pixels=None, batch_size=None):
    """How many steps of an epoch a lone fit takes on the half-chip launch, beside the previous epoch's background
    evaluation pass (fit_device): as long as that pass takes.  Both durations follow from the shape's FLOP counts and the
    achieved fractions above -- the pass 2 x N x forward / (0.65 peak), a step B x (forward + backward) / (0.26 peak) + 5 us:
    2.9 ms against 20.4 us at the headline shape = 0.27 of an epoch's 512 steps, what the sweep over the fraction also found
    (0.15 / 0.2 / 0.27 / 0.35 / 0.45: 106.4 / 105.5 / 104.3 / 105.7 / 106.5 ms per tile).  The ratio does not depend on the image
    size.  LBDRN_LONE_HEAD_FRAC overrides it (A/B)."""
    env = os.environ.get("LBDRN_LONE_HEAD_FRAC")
    if env is not None:
        return int(round(float(env) * steps_per_epoch))
    if net is None:
        return int(round(0.27 * steps_per_epoch))
    F, bc, C, nl = int(net.F), int(net.bc), int(net.C), int(net.nl)
    fwd = 2 * (F * bc + (nl - 1) * bc * bc + bc * C)
    step = 2 * fwd + 2 * ((nl - 1) * bc * bc + bc * C)       # forward + weight gradients + input gradients of the hidden layers
    t_pass = 2.0 * n_pixels * fwd / (EVAL_PASS_FRAC_OF_PEAK * PEAK_FLOPS)
    t_step = min(batch_size, n_pixels) * step / (HALF_CHIP_STEP_FRAC_OF_PEAK * PEAK_FLOPS) + HALF_CHIP_STEP_OVERHEAD_S
    return max(0, min(steps_per_epoch, int(round(t_pass / t_step))))


def _eval_stream(dev, main):
    """The stream a lone fit's evaluation passes run on: one of the fit streams (none is busy when a fit is alone),
    not a stream of its own -- a process that uses more streams than the device has hardware queues makes streams
    share a queue, and a background pass that shares one with the training chain costs 3x the fit instead of
    saving 5 % of it (measured: scripts/ab_streamidx.py)."""
    for s in _fit_stream_pool(dev, 2):
        if s.cuda_stream != main.cuda_stream:
            return s


def fit_device(img_d, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration=1,
               cfg=None, path=ops.PATH_AUTO, keep_losses=False, seed=None, draws=None, alone=True):
    """Fit one image that already sits in HBM (img_d: [C,H,W] uint16 bits in int16 storage), on the
    calling thread's current stream.

    The caller has seeded torch (ref encode.py:200-205); the global CPU generator is consumed
    exactly as the reference's train() consumes it: model construction, then one DataLoader
    iterator per train / eval pass (lbdrn_hip.sampler).  With `seed`, the fit seeds the generator
    itself and makes all its draws in one critical section, so that fits running on several
    threads (fit_many) each see what a freshly seeded process would; with `draws` (draw_fit) the generator
    is not touched at all.  `alone`: no other fit is in flight in this process -- its evaluation passes then run
    in the background of the next epoch's training where overlap_evaluation() allows it (below).  No host
    synchronisation happens inside except the scalar read of MSB.max() that sizes the normalisation
    (ref LBDRNdataset.py:120)."""
    cfg = cfg or FeatCfg.from_constants()
    out = DeviceFit()
    dev = img_d.device
    C, H, W = img_d.shape
    N = H * W
    alone = alone and not device_shared()
    msb_d, msb_max = ops.split_bits(img_d, K)            # a1
    geom = ops.FeatureGeometry(C, H, W, K, D, msb_max, cfg, dev)
    if N >= GPU_RANDPERM_MAX:
        raise ops._lib.LbdrnError(
            f"{N} pixels in one fit: torch.randperm switches algorithm at 2^32/20 elements and lbdrn_randperm "
            "implements the Fisher-Yates branch only; split the image (-sr) -- there is no host fallback")
    if draws is None:
        with _RNG_LOCK:
            if seed is not None:
                torch.manual_seed(seed)
            # model on the CPU first, like the reference (encode.py:71-77): consumes the global generator;
            # then every pass's sampler seed
            draws = draw_fit(geom.F, base_channel, C, num_layers, epochs, val_duration)
    stream = DevicePermutationStream(N, epochs, val_duration, dev, train_seeds=draws.train_seeds)
    net = ops.make_net(geom.F, base_channel, C, num_layers, cfg.act)
    params = draws.params.to(dev).contiguous()
    if params.numel() != ops.param_count(net):
        raise ValueError("draws were made for another network shape")
    draws_params0 = params.clone()
    exp_avg = torch.zeros_like(params)
    exp_avg_sq = torch.zeros_like(params)
    lrs = lr_schedule(lr, epochs)
    steps_per_epoch = (N + batch_size - 1) // batch_size
    losses = torch.zeros((epochs, steps_per_epoch), dtype=torch.float32, device=dev) if keep_losses else None
    train_ws = ops.TrainWorkspace(geom, net, batch_size, dev).prepare(img_d, msb_d, path)   # a2-a4
    apply_ws = ops.ApplyWorkspace(geom, net, dev)
    mse_log = torch.zeros((epochs, 2), dtype=torch.float32, device=dev)
    eval_epochs = [] if epochs == 1 else [e for e in range(1, epochs + 1) if e % min(val_duration, epochs) == 0]
    # A fit alone on the device: its train steps are a chain of short kernels on half of the CUs, so the evaluation
    # pass of epoch e (a9) reads a snapshot of the weights on a second stream, launched on half as many workgroups
    # (EVAL_BACKGROUND: same sum bit for bit), while epoch e+1 trains on the other half -- instead of standing in
    # the chain.  With other fits in flight the chip is full anyway and a whole-chip pass in the chain is faster
    # (measured).  Which epoch was best is settled after the last pass: same rule, same order.
    main = torch.cuda.current_stream(dev)
    side = _eval_stream(dev, main) if len(eval_epochs) > 1 and overlap_evaluation(dev, alone) else None
    snaps = torch.empty((max(len(eval_epochs), 1), params.numel()), dtype=torch.float32, device=dev)
    mses = torch.zeros((max(len(eval_epochs), 1),), dtype=torch.float32, device=dev)
    if side is not None:
        side.wait_stream(main)       # the workspaces and the planes above are ready
    adam_steps = 0
    # (the first background pass and the steps beside it are timed with events nobody waits for: _calibrated_head)
    hkey = _head_key(dev, net, N, batch_size)
    cal_epoch = eval_epochs[0] if eval_epochs else 0
    cal_events = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if (side is not None and alone and cal_epoch + 1 <= epochs) else None
    try:
        for e in range(1, epochs + 1):
            perm = stream.get(e).to(dev, non_blocking=True)                  # a4 (already on the device)
            # While the previous epoch's evaluation pass runs in the background it holds half of the CUs: the steps beside
            # it go out WITHOUT the alone hint -- the half-chip launch that fits on the other half (k_train_stream) --, the
            # rest of the epoch with it (k_train_split, every CU).  Same numbers either way (lbdrn_hip.h: the hint
            # changes no bit), so where the line is drawn is a matter of time only.
            head = 0
            if side is not None and alone and e - 1 in eval_epochs:
                head = background_steps(steps_per_epoch, net, N, batch_size)
                if "LBDRN_LONE_HEAD_FRAC" not in os.environ:
                    head = _calibrated_head(hkey, head, steps_per_epoch)
            for lo, hi, hint in ((0, head, False), (head, steps_per_epoch, alone)):
                if hi > lo:
                    timed = cal_events is not None and lo == 0 and head > 0 and e == cal_epoch + 1
                    if timed:
                        cal_events[2].record(main)
                    ops.train_epoch(geom, net, img_d, msb_d, perm[lo * batch_size:hi * batch_size], batch_size, params, exp_avg,
                                    exp_avg_sq, adam_steps + lo, lrs[e - 1], losses[e - 1][lo:hi] if keep_losses else None, path,
                                    train_ws, alone=hint)
                    if timed:
                        cal_events[3].record(main)
                        with _HEAD_LOCK:
                            _HEAD_PENDING[hkey] = (cal_events, head)
            adam_steps += steps_per_epoch
            if e in eval_epochs:                                              # encode.py:104-117
                k = eval_epochs.index(e)
                snaps[k].copy_(params)
                if side is not None:
                    side.wait_stream(main)
                with torch.cuda.stream(side if side is not None else main):
                    background = side is not None and e != epochs      # nothing trains beside the last pass
                    if cal_events is not None and e == cal_epoch:
                        cal_events[0].record()
                    sse = ops.eval_sse(geom, net, img_d, msb_d, snaps[k], path, apply_ws, background=background,
                                       fast=fast_evaluation(), x16=exact16_evaluation())   # a9
                    if cal_events is not None and e == cal_epoch:
                        cal_events[1].record()
                    mses[k:k + 1].copy_((sse / float(N * C)).float())
                out.evaluated.append(e)
    finally:
        if side is not None:       # whatever happened, the borrowed stream is joined before anybody else uses it
            main.wait_stream(side)
    best_params = params.clone() if epochs == 1 else draws_params0      # encode.py:100-103 / :91
    best_mse = torch.full((1,), 1e6, dtype=torch.float32, device=dev)   # encode.py:91
    for k, e in enumerate(eval_epochs):
        mse = mses[k:k + 1]
        improved = mse < best_mse
        best_params = torch.where(improved, snaps[k], best_params)
        best_mse = torch.where(improved, mse, best_mse)
        mse_log[e - 1, 0] = mse[0]
        mse_log[e - 1, 1] = improved[0].float()
    stream.close()
    out.best_params, out.msb, out.msb_max, out.geom, out.net = best_params, msb_d, msb_max, geom, net
    out.mse_log, out.losses, out.epochs = mse_log, losses, epochs
    return out


def fit_group(imgs_d, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration=1, cfg=None,
              path=ops.PATH_AUTO, keep_losses=False, seed=None, draws=None):
    """Several independent fits of ONE raster shape, stepping side by side on the calling thread's stream: every
    minibatch of the group is one launch (ops.train_epoch_group) -- two 8192-row minibatches are 256 workgroups,
    the whole chip, where two independent chains of 128-workgroup launches only meet by chance.  Each fit is exactly
    fit_device()'s fit (same draws, same kernels per fit, own workspace and state): the results are bit-identical to
    fitting the images one after another.  `seed`: every fit seeds the generator itself (independent images, one
    encode.py invocation each); `draws`: one FitDraws per image.  Returns one DeviceFit per image."""
    cfg = cfg or FeatCfg.from_constants()
    G = len(imgs_d)
    dev = imgs_d[0].device
    C, H, W = imgs_d[0].shape
    if any(tuple(t.shape) != (C, H, W) for t in imgs_d):
        raise ValueError("the fits of a group must have one raster shape")
    N = H * W
    if N >= GPU_RANDPERM_MAX:
        raise ops._lib.LbdrnError(f"{N} pixels in one fit: split the image (-sr)")
    outs = [DeviceFit() for _ in range(G)]
    st = []
    net = None
    for k, img_d in enumerate(imgs_d):
        msb_d, msb_max = ops.split_bits(img_d, K)            # a1
        geom = ops.FeatureGeometry(C, H, W, K, D, msb_max, cfg, dev)
        dr = draws[k] if draws is not None else None
        if dr is None:
            with _RNG_LOCK:
                if seed is not None:
                    torch.manual_seed(seed)
                dr = draw_fit(geom.F, base_channel, C, num_layers, epochs, val_duration)
        net = ops.make_net(geom.F, base_channel, C, num_layers, cfg.act)
        params = dr.params.to(dev).contiguous()
        if params.numel() != ops.param_count(net):
            raise ValueError("draws were made for another network shape")
        steps = (N + batch_size - 1) // batch_size
        st.append(dict(
            img=img_d, msb=msb_d, msb_max=msb_max, geom=geom, params=params, params0=params.clone(),
            m=torch.zeros_like(params), v=torch.zeros_like(params),
            perms=DevicePermutationStream(N, epochs, val_duration, dev, train_seeds=dr.train_seeds),
            losses=torch.zeros((epochs, steps), dtype=torch.float32, device=dev) if keep_losses else None,
            tws=ops.TrainWorkspace(geom, net, batch_size, dev).prepare(img_d, msb_d, path),   # a2-a4
            aws=ops.ApplyWorkspace(geom, net, dev)))
    lrs = lr_schedule(lr, epochs)
    steps_per_epoch = (N + batch_size - 1) // batch_size
    eval_epochs = [] if epochs == 1 else [e for e in range(1, epochs + 1) if e % min(val_duration, epochs) == 0]
    for f in st:
        f["snaps"] = torch.empty((max(len(eval_epochs), 1), f["params"].numel()), dtype=torch.float32, device=dev)
        f["mses"] = torch.zeros((max(len(eval_epochs), 1),), dtype=torch.float32, device=dev)
    adam_steps = 0
    for e in range(1, epochs + 1):
        perms = [f["perms"].get(e).to(dev, non_blocking=True) for f in st]          # a4
        ops.train_epoch_group([f["geom"] for f in st], net, [f["img"] for f in st], [f["msb"] for f in st], perms,
                              batch_size, [f["params"] for f in st], [f["m"] for f in st], [f["v"] for f in st],
                              adam_steps, lrs[e - 1], [f["losses"][e - 1] for f in st] if keep_losses else None,
                              path, [f["tws"] for f in st])
        adam_steps += steps_per_epoch
        if e in eval_epochs:                                                          # encode.py:104-117
            k = eval_epochs.index(e)
            for f, out in zip(st, outs):
                f["snaps"][k].copy_(f["params"])
                sse = ops.eval_sse(f["geom"], net, f["img"], f["msb"], f["snaps"][k], path, f["aws"], fast=fast_evaluation(), x16=exact16_evaluation())   # a9
                f["mses"][k:k + 1].copy_((sse / float(N * C)).float())
                out.evaluated.append(e)
    for f, out in zip(st, outs):
        best_params = f["params"].clone() if epochs == 1 else f["params0"]      # encode.py:100-103 / :91
        best_mse = torch.full((1,), 1e6, dtype=torch.float32, device=dev)       # encode.py:91
        mse_log = torch.zeros((epochs, 2), dtype=torch.float32, device=dev)
        for k, e in enumerate(eval_epochs):
            mse = f["mses"][k:k + 1]
            improved = mse < best_mse
            best_params = torch.where(improved, f["snaps"][k], best_params)
            best_mse = torch.where(improved, mse, best_mse)
            mse_log[e - 1, 0] = mse[0]
            mse_log[e - 1, 1] = improved[0].float()
        f["perms"].close()
        out.best_params, out.msb, out.msb_max, out.geom, out.net = best_params, f["msb"], f["msb_max"], f["geom"], net
        out.mse_log, out.losses, out.epochs = mse_log, f["losses"], epochs
    return outs


def fit_many(images, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration=1, cfg=None,
             path=ops.PATH_AUTO, seed=19920517, in_flight=None, then=None, draws=None, group=None):
    """Fit several HBM-resident images on ONE GPU with `in_flight` of them progressing at a time (None: default_in_flight of
    the shape -- 4, or 3 at bc >= 128 --, cut to what the free memory holds); returns
    [then(fit) or fit, ...] in input order.

    Why: one fit is a strict chain of short dependent kernels (train step ~16 us on half the chip -> reduce/Adam
    ~5 us -> train ...), so several independent fits progress together.  How they share the chip is `group`:
    fits of one raster shape are taken `group` at a time and step side by side in ONE launch per minibatch
    (fit_group: 2 x 128 workgroups = every CU, deterministically), and in_flight // group such groups run on their
    own streams and host threads, one training while the other reduces.  group=None: LBDRN_FIT_GROUP, else 2 with four
    or more in flight, else 1.  What decides it is the number of kernel boundaries on the chip: every launch boundary is
    an L2 write-back + invalidate across the eight XCDs, and with several chains in flight the boundaries of one chain
    lengthen those of the others (in-kernel timeline of a step, scripts/stamp_probe_inflight.py on the -DLBDRN_TIMELINE
    build: 20.8 us alone, 26 with two chains, 34-46 with four) -- pairs halve them.  Measured on 8 x 2048^2 tiles, ms
    per tile at group : in flight -- 1:4 69.2, 2:4 64.7, 2:6 66.9, 3:6 66.7, 3:9 66.0, 4:8 66.0, 4:4 77.8, 2:2 87.0,
    1:3 70.1 (before the evaluation pass and the reduce launch were made cheaper, round 3's first measurements, pairs
    lost: 1:4 72.8, 2:4 75.5).  Images are independent fits and
    every fit seeds the generator itself (`seed`, what each encode.py invocation does, ref encode.py:200-205), so
    results are bit-identical to fitting them one after another, whatever the grouping.
    `then(fit)` runs on the worker's stream right after its fit (weight truncation + reconstruction, payload
    coding, ...).  `draws`: one FitDraws per image instead of `seed` (the tiles of one image, whose draws the
    caller made in tile order).  Returns after all streams have been joined to the caller's current stream."""
    if in_flight is None:
        in_flight = default_in_flight(*tuple(images[0].shape), K, D, base_channel, num_layers, cfg, path) if images else 4
    in_flight = memory_limited_in_flight(images, in_flight, K, D, base_channel, num_layers, batch_size, epochs, cfg)
    if group is None:
        group = int(os.environ.get("LBDRN_FIT_GROUP", "0"))
        if group == 0:
            group = 1
            if in_flight >= 4 and images:
                C, H, W = images[0].shape
                fcfg = cfg or FeatCfg.from_constants()
                if ops.train_group_size(C, H, W, K, D, fcfg, base_channel, num_layers) >= 2 and path != ops._lib.PATH_GENERIC:
                    group = 2
    group = max(1, min(group, in_flight, ops.train_group_max()))
    if draws is not None:
        seed = None
        if len(draws) != len(images):
            raise ValueError("one FitDraws per image expected")
    elif seed is None and in_flight > 1 and len(images) > 1:
        raise ValueError("fits in flight together must seed themselves (seed=...) or bring their draws: "
                         "they share one generator")
    caller = torch.cuda.current_stream(images[0].device) if images else None
    local = threading.local()
    taken = []   # worker index -> stream of the process-wide pool

    def work(job):
        idx, imgs, drs = job
        if not hasattr(local, "stream"):
            # streams are kept for the life of the process: torch's caching allocator pools memory per stream,
            # so a fresh stream per call would send every fit's 3.5 GB workspace back to hipMalloc
            with _POOL_LOCK:
                k = len(taken)
                taken.append(None)
            local.stream = _fit_stream_pool(imgs[0].device, k + 1)[k]
            local.stream.wait_stream(caller)
        with torch.cuda.stream(local.stream):
            if len(imgs) == 1:
                fits = [fit_device(imgs[0], K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration, cfg,
                                   path, seed=seed, draws=drs[0], alone=False)]
            else:
                fits = fit_group(imgs, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration, cfg, path,
                                 seed=seed, draws=drs if draws is not None else None)
            outs = [then(fit) if then is not None else fit for fit in fits]
            done = torch.cuda.Event()
            done.record(local.stream)
        return idx, outs, done

    drs_all = draws if draws is not None else [None] * len(images)
    if in_flight <= 1 or len(images) <= 1:
        results = []
        for img_d, dr in zip(images, drs_all):
            fit = fit_device(img_d, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration, cfg, path,
                             seed=seed, draws=dr)
            results.append(then(fit) if then is not None else fit)
        return results
    # consecutive images of one shape, `group` at a time
    def form(group):
        jobs, k = [], 0
        while k < len(images):
            n = 1
            while n < group and k + n < len(images) and tuple(images[k + n].shape) == tuple(images[k].shape):
                n += 1
            jobs.append((list(range(k, k + n)), list(images[k:k + n]), list(drs_all[k:k + n])))
            k += n
        return jobs
    jobs = form(group)
    if group > 1 and 2 * sum(len(j[0]) for j in jobs if len(j[0]) > 1) < len(images):
        # mostly odd ones out (tiles of many shapes): groups would only halve the number of chains
        group = 1
        jobs = form(1)
    with ThreadPoolExecutor(max_workers=max(1, in_flight // group)) as pool:
        done_jobs = list(pool.map(work, jobs))
    results = [None] * len(images)
    for idx, outs, done in done_jobs:
        caller.wait_event(done)
        for i, o in zip(idx, outs):
            results[i] = o
    return results


def truncate_device(params, precision):
    """Device-side twin of container.truncate_precision (what the decoder sees after the weight
    payload round trip)."""
    if precision in (0, 32):
        return params.clone()
    mask = -(1 << (32 - precision))
    return (params.view(torch.int32) & mask).view(torch.float32)


def apply_device(geom, net, msb_d, params_d, path=ops.PATH_AUTO, ws=None, want_y=False):
    """Reconstruct from HBM-resident MSB plane + weights (ref decode.py:122-134)."""
    return ops.decode_fused(geom, net, msb_d, params_d, want_y=want_y, path=path, ws=ws)


def _host_result(fit, epochs, seconds, host_msb):
    """DeviceFit -> FitResult (one host sync: the evaluation log)."""
    res = FitResult()
    res.seconds = dict(seconds)
    mse_host = fit.mse_log.cpu().numpy()
    if epochs == 1:
        res.best_epoch = 1
    for e in fit.evaluated:
        res.epoch_mse.append((e, float(mse_host[e - 1, 0]), bool(mse_host[e - 1, 1] > 0)))
        if mse_host[e - 1, 1] > 0:
            res.best_epoch, res.best_mse = e, float(mse_host[e - 1, 0])
    if epochs > 1 and fit.evaluated and res.best_epoch < 0:
        # the reference would stop at torch.load of a model.pt that was never saved (ref encode.py:108-122);
        # writing the initial weights into a bitstream silently is worse than stopping
        worst = [m for _, m, _ in res.epoch_mse]
        raise ops._lib.LbdrnError(
            f"no evaluation pass improved on the initial best MSE of 1e6 (per-epoch MSE {worst}): the fit diverged or "
            "the input is degenerate (e.g. an all-zero MSB plane gives 0/0 features); no bitstream is written")
    res.params = fit.best_params.cpu().numpy()
    res.msb_device = fit.msb
    if host_msb:
        msb = ops.from_device_u16(fit.msb)
        res.msb = msb.astype(np.uint16) if fit.msb_max > 255 else msb.astype(np.uint8)   # LBDRNdataset.py:100
    res.msb_max = fit.msb_max
    C, H, W = fit.msb.shape
    res.n_feature, res.channels, res.n_subpixels = fit.geom.F, C, H * W * C
    res.losses = fit.losses
    return res


def _as_planes(img):
    img = np.ascontiguousarray(img, dtype=np.uint16)
    return img[None] if img.ndim == 2 else img


def fit_image(img, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration=1, cfg=None,
              device="cuda:0", path=ops.PATH_AUTO, keep_losses=False, host_msb=True, draws=None):
    """Host-array front end of fit_device(): img numpy uint16 [C,H,W] or [H,W] -> FitResult.
    host_msb=False leaves the MSB plane in HBM only (the encoder codes it there)."""
    t0 = time.time()
    img_d = ops.to_device_u16(_as_planes(img), torch.device(device))
    t1 = time.time()
    fit = fit_device(img_d, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration, cfg,
                     path, keep_losses, draws=draws)
    res = _host_result(fit, epochs, {"upload": t1 - t0}, host_msb)
    res.seconds["fit"] = time.time() - t1
    return res


def fit_images(imgs, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration=1, cfg=None,
               device="cuda:0", path=ops.PATH_AUTO, host_msb=True, draws=None, seed=None, in_flight=None):
    """fit_image for several rasters with `in_flight` of them progressing at a time on the GPU (fit_many).
    Either every fit seeds itself (`seed`: independent images, one encode.py invocation each in the reference)
    or the caller brings the draws it made in order (`draws`: the tiles of one image, ref encode.py:231-262)."""
    t0 = time.time()
    dev = torch.device(device)
    tiles = [ops.to_device_u16(_as_planes(img), dev) for img in imgs]
    t1 = time.time()
    fits = fit_many(tiles, K, D, base_channel, num_layers, lr, batch_size, epochs, val_duration, cfg, path,
                    seed=seed, in_flight=in_flight, draws=draws)
    results = [_host_result(fit, epochs, {"upload": (t1 - t0) / max(len(imgs), 1)}, host_msb) for fit in fits]
    per_tile = (time.time() - t1) / max(len(imgs), 1)
    for res in results:
        res.seconds["fit"] = per_tile          # wall time per tile with the others in flight
    return results


def apply_image(base, params, K, D, base_channel, num_layers, cfg=None, device="cuda:0",
                path=ops.PATH_AUTO, want_y=False):
    """Reconstruct one image (or tile) from its MSB plane and fitted weights (ref decode.py:73-134).
    base: numpy [C,H,W] (uint8 or uint16), or the plane already in HBM (int16-storage tensor);
    params: float32 vector in state_dict order."""
    cfg = cfg or FeatCfg.from_constants()
    dev = torch.device(device)
    if isinstance(base, torch.Tensor):
        msb_d = base if base.dim() == 3 else base[None]
        # uint16 bits in int16 storage: the maximum as unsigned
        msb_max = int((msb_d.to(torch.int32) & 0xFFFF).max().item())
    else:
        base = np.ascontiguousarray(base).astype(np.uint16)               # decode.py:74
        if base.ndim == 2:
            base = base[None]
        msb_d = ops.to_device_u16(base, dev)
        msb_max = int(base.max())
    C, H, W = msb_d.shape
    geom = ops.FeatureGeometry(C, H, W, K, D, msb_max, cfg, dev)         # divisor base.max(): decode.py:93
    net = ops.make_net(geom.F, base_channel, C, num_layers, cfg.act)
    p = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float32)).to(dev)
    if p.numel() != ops.param_count(net):
        raise ValueError(f"weight payload has {p.numel()} values, the network needs {ops.param_count(net)}")
    out = ops.decode_fused(geom, net, msb_d, p, want_y=want_y, path=path)
    if want_y:
        return ops.from_device_u16(out[0]), out[1].cpu().numpy()
    return ops.from_device_u16(out)
