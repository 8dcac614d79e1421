"""Per-shape algorithm selection by measurement ("measure, don't guess"): the caches ops.py consults, the candidate lists and
the timing loops.  `with tuning(): ...` makes every conv / weight-gradient shape met for the first time time all of its
candidates on the device and cache the winner; without it an uncached shape runs the library's default (algo 0).
dist_utils.broadcast_tuner_choices / save_tuner_choices / load_tuner_choices move the three caches between ranks and runs."""
import os

import torch

from . import _lib
from ._lib import OSD_BF16

class _Cache(dict):
    """a tuner cache: a dict that can count its lookups per key (`census`) — how often a step launches each tuned shape, which
    is what refine_in_step weighs a choice by"""
    census = False

    def __init__(self, *a, **kw):
        dict.__init__(self, *a, **kw)
        self.hits = {}

    def get(self, key, default=None):
        if self.census:
            self.hits[key] = self.hits.get(key, 0) + 1
        return dict.get(self, key, default)


SPLIT_CACHE = _Cache()          # grouped launches: where to cut the segment list into a large-level and a small-level launch
WGRAD_ALGO_CACHE = _Cache()     # weight-gradient launches: 1 + variant + 16 * split-target code
TUNE_LOG = {"ALGO_CACHE": {}, "WGRAD_ALGO_CACHE": {}, "SPLIT_CACHE": {}}      # cache name -> key -> sorted [(ms, choice)]: what the isolated timing saw

# osd_conv_desc.algo = 1 + impl*32 + variant*8 + tile; impl 0 = LDS-DMA ring kernel (variants: deep / shallow ring /
# short stages), impl 1 = register-staged kernel; tile 0..4 = 128x128, 128x64, 64x64, 256x16, 256x256/8 waves, 5 = 128x256/8 waves
# (round 5; the ping-pong 256x256 kernel of rounds 1-4 is retired), 6 = row-reuse 3x3 conv_sp (variant 0: 128 x 128 tiles, round 5 — the id of the
# retired conv_xr; 1: any width; 2: the general-width form forced; 3: 128-pixel tiles), 7 = 256x128/8 waves (pixels x channels).
ALGO_CACHE = _Cache()
_TUNING = [False]


CONV_ALGO_DEEP5 = 1 + 1 * 32 + 3 * 8 + 1   # the 64 x 64 LDS-DMA tile with a ring of five stages (round 6)
CONV_ALGO_DEEP8 = 1 + 1 * 32 + 3 * 8 + 2   # ... eight stages: 112 KB of operands in flight per workgroup
COLD_MAX_PIXELS = int(os.environ.get("OSD_TUNE_COLD_PIXELS", "32768"))      # convs with at most this many output pixels are timed with cold weights (below)
DEEP_MAX_PIXELS = 4096                      # ... and get the deep-ring 64 x 64 tiles as candidates
CONV_ALGO_PRED = 1 + 1 * 32 + 2 * 8 + 2    # conv_pred.hip (round 6): 3x3 convs with <= 4 output channels on an 8 x 32 output patch per workgroup
CONV_ALGO_PX = 1 + 1 * 32 + 2 * 8 + 0      # conv_px.hip, eight waves of 16 pixels
CONV_ALGO_PX_WIDE = CONV_ALGO_PX + 1       # ... four waves of 32 pixels


def conv_algo_candidates(cout_store, relu_in, has_mask=False, pixels=None):
    tiles = [3] if cout_store <= 16 else [0, 1, 2]
    if cout_store <= 16:
        tiles = [3, 2]
    cands = [1 + 0 * 32 + v * 8 + t for v in (0, 1, 2, 3) for t in tiles]      # v 3: three short stages (many workgroups per CU)
    if cout_store <= 4 and not relu_in and not has_mask and not os.environ.get("OSD_NO_PRED_KERNEL"):
        cands.append(CONV_ALGO_PRED)                  # the FCOS prediction convs' forward (bf16, 3x3 / 1 / 1; refused elsewhere)
    if cout_store >= 256 and not relu_in:
        cands.append(1 + 0 * 32 + 1 * 8 + 4)          # 256x256 tile, shallow ring
        cands.append(1 + 0 * 32 + 2 * 8 + 4)          # 256x256 tile, short stages x 4
        if not os.environ.get("OSD_NO_SP"):           # (A/B switch for tools and benches)
            cands.append(1 + 0 * 32 + 1 * 8 + 6)      # 3x3/1: pixel rows fetched once per filter row, software-pipelined fragments + mid-stage barrier, ANY width (bf16)
            cands.append(1 + 0 * 32 + 3 * 8 + 6)      # ... on 128-pixel tiles: twice the workgroups where 256-pixel tiles leave CUs idle
            cands.append(1 + 0 * 32 + 0 * 8 + 6)      # ... on 128 x 128 tiles: half the weight stream per workgroup (layer4's 3x3: K = 4,608 on 6,400 pixels)
    if 64 < cout_store <= 128 and not relu_in and not os.environ.get("OSD_NO_SP") and not os.environ.get("OSD_NO_SP_NARROW"):
        cands.append(1 + 0 * 32 + 1 * 8 + 6)          # the same kernel on a 256-pixel x 128-channel tile (4 x 2 waves): layer2's 3x3 convs
        if not os.environ.get("OSD_NO_SP_SMALL128"):
            cands.append(1 + 0 * 32 + 0 * 8 + 6)      # ... and on the 128 x 128 tile: 76 KB of LDS, two workgroups per CU (round 6)
    if cout_store >= 128 and not relu_in:
        cands += [1 + 0 * 32 + v * 8 + 7 for v in (0, 1, 2, 3)]      # 256x128 tile on 8 waves: deep / shallow ring / short stages
    if cout_store >= 256 and not relu_in:
        cands += [1 + 0 * 32 + v * 8 + 5 for v in (0, 1, 2, 3)]      # 128x256 tile on 8 waves: all of N = 256 per pixel tile (reducing 1x1 convs)
    if not relu_in and not has_mask:
        cands += [1 + 1 * 32 + t for t in tiles]
    if cout_store > 16 and not relu_in and pixels is not None and pixels <= DEEP_MAX_PIXELS and not os.environ.get("OSD_NO_DEEP_RING"):
        # latency-sized launches (the query backbone at bs 8): 16 - 64 workgroups each streaming a slice of a weight matrix that is
        # cold inside the step — deep rings keep 64 / 112 KB in flight per workgroup (bf16, cin in 64s; refused elsewhere)
        cands += [CONV_ALGO_DEEP5, CONV_ALGO_DEEP8, CONV_ALGO_DEEP8 - 2, CONV_ALGO_DEEP8 + 1]      # + 64 x 32 and 32 x 64 tiles, eight stages (57, 60)
    if cout_store >= 64 and not relu_in and os.environ.get("OSD_PX"):
        # pixel-stationary pointwise kernel (round 5; bf16 1x1 / stride 1 convs with cin 64 / 128 / 256: refused elsewhere).  Opt-in,
        # like round 4's persistent conv_pw (retired): timed alone it wins layer2's expanding convs by 8 - 10 % and the tuner picks it there, but inside the step
        # its long-lived workgroups share the chip no better than the tile kernels' short ones: 699 - 715 images/s with it, 711 -
        # 714 without, 692 - 699 for the round-4 head on the same box (profiles/r5_same_box_ab.txt)
        cands += [CONV_ALGO_PX, CONV_ALGO_PX_WIDE]
    return cands


class tuning(object):
    """with ops.tuning(): ...   every conv shape met for the first time is timed over all candidate algorithms."""

    def __enter__(self):
        _TUNING[0] = True

    def __exit__(self, *a):
        _TUNING[0] = False


class replaying(object):
    """with ops.replaying(): ...   the ranks other than the tuning one run a step from the caches they were sent
    (dist_utils.broadcast_tuner_choices): every shape must hit a cache.  A miss — a shape only this rank meets — raises at the
    point it occurs instead of being timed silently on this rank alone and noticed later by tuner_choices_agree."""

    def __enter__(self):
        _TUNING[0] = True
        _REPLAY[0] = True

    def __exit__(self, *a):
        _TUNING[0] = False
        _REPLAY[0] = False


_REPLAY = [False]


def _time_launches(fn):
    """Milliseconds per launch of `fn`, back to back on the current stream: OSD_TUNE_REPS launches per bracket (default 3), the
    fastest of OSD_TUNE_ROUNDS brackets (default 1)."""
    if _REPLAY[0]:
        raise RuntimeError("tuner cache miss while replaying another rank's choices: this rank met a shape the tuning rank did not")
    reps, rounds = int(os.environ.get("OSD_TUNE_REPS", "3")), int(os.environ.get("OSD_TUNE_ROUNDS", "1"))
    best = float("inf")
    for _ in range(rounds):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) / reps)
    return best


_FLUSH = [None]


def _time_launches_cold(fn, prewarm=None):
    """Milliseconds per launch of `fn` with COLD caches: before every timed launch a 512 MB elementwise pass evicts the L2s and the
    256 MB Infinity Cache, then `prewarm()` (optional) touches what the launch's producer would have left on chip (its activations).
    For the latency-sized convs of the query backbone: inside a training step a weight matrix is read once per step with ~1 GB of
    parameter state in between, so it comes from HBM — and the ring depth that wins a warm back-to-back loop (two stages: 14 us warm,
    31 us cold at 512 x 256 x 2,304) is the slowest cold (deep rings: 20 - 23 us; tools/small_m_cold.py)."""
    if _REPLAY[0]:
        raise RuntimeError("tuner cache miss while replaying another rank's choices: this rank met a shape the tuning rank did not")
    if _FLUSH[0] is None:
        _FLUSH[0] = torch.zeros(128 * 1024 * 1024, device="cuda")
    reps = max(3, int(os.environ.get("OSD_TUNE_REPS", "3")))
    total = 0.0
    for _ in range(reps):
        _FLUSH[0].add_(1.0)
        if prewarm is not None:
            prewarm()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        fn()
        ev[1].record()
        torch.cuda.synchronize()
        total += ev[0].elapsed_time(ev[1])
    return total / reps


def release_tuning_scratch():
    """free the cold-timing flush buffer (512 MB) once tuning is over"""
    _FLUSH[0] = None


def _photo_finish(timed, retime):
    """timed: [(ms, algo)] of one bracket each.  Candidates within OSD_TUNE_MARGIN (default 5 %) of the fastest — at most three — are
    timed twice more and the fastest of a candidate's brackets decides: one three-launch bracket carries a few per cent of noise,
    and a wrong pick on a 0.6 ms weight-gradient launch is a per cent of the step for the whole run."""
    timed = sorted(timed)
    if not timed:
        return 0
    margin = 1.0 + float(os.environ.get("OSD_TUNE_MARGIN", "0.05"))
    close = [(t, a) for t, a in timed if t <= timed[0][0] * margin][:3]
    if len(close) < 2:
        return timed[0][1]
    best = {a: t for t, a in close}
    for _ in range(2):
        for _, a in close:
            best[a] = min(best[a], retime(a))
    return min(best.items(), key=lambda kv: kv[1])[0]


def _tune(key, d, launch, cands=None, prewarm=None):
    """Time every candidate algorithm for this conv shape (launch() reads d.algo) and cache the fastest.  Shapes with at most
    COLD_MAX_PIXELS output pixels are timed with cold caches (`prewarm` re-touches the activations): see _time_launches_cold."""
    timed = []
    pixels = d.n * d.ho * d.wo
    # (grouped / multi launches leave d.n = 0: several tensors; their weights were used a launch earlier — warm)
    cold = 0 < pixels <= COLD_MAX_PIXELS and d.cout > 16 and not os.environ.get("OSD_TUNE_WARM_ONLY")
    time_it = (lambda f: _time_launches_cold(f, prewarm)) if cold else _time_launches
    forced = os.environ.get("OSD_FORCE_ALGO_3X3_256")     # experiments: pin the big 3x3 256-channel convs to one algorithm
    if forced and d.cout == 256 and d.r == 3 and d.stride_h == 1:
        d.algo = int(forced)
        try:
            launch()
            torch.cuda.synchronize()
            ALGO_CACHE[key] = int(forced)
            return int(forced)
        except _lib.OsdError:
            pass
    for algo in (cands if cands is not None else conv_algo_candidates(d.cout, d.relu_in, has_mask=bool(key[-1]), pixels=pixels if pixels > 0 else None)):
        d.algo = algo
        try:
            launch()
        except _lib.OsdError:
            continue
        timed.append((time_it(launch), algo))

    def retime(algo):
        d.algo = algo
        return time_it(launch)
    best = _photo_finish(timed, retime)
    ALGO_CACHE[key] = best
    TUNE_LOG["ALGO_CACHE"][key] = sorted(timed)
    return best


def wgrad_xr_candidates(dtype, cout, cin, r, s, stride, pad, widths):
    """The filter-row kernel (one workgroup per 3 taps; conv_wgrad_xr_kernel): algo = 1 + 128 + x + 16 * split-target code,
    x = 0: 32-pixel stages x 6, 1: 64 x 4, 2: 32 x 8.  bf16 3x3 / 1 / 1, channels in 128s, every map width a multiple or
    a divisor of the stage."""
    if dtype != OSD_BF16 or (r, s, stride, pad) != (3, 3, 1, 1) or cout % 128 or cin % 128 or not widths:
        return []
    xs = [x for x, bk in ((0, 32), (1, 64), (2, 32)) if all(w % bk == 0 or bk % w == 0 for w in widths)]
    return [1 + 128 + x + 16 * t for t in (0, 1, 2, 4, 5, 6, 7) for x in xs]


def wgrad_algo_candidates(dtype, cout=0, cin=0):
    """osd_conv2d_wgrad's algo field = 1 + variant + 16 * split-target code.  Variants 0..3: 128 x 128 channel tile with
    different stage shapes (bf16; fp32 has one); 4..9: 256-wide channel tiles on 8 waves (bf16, wide layers; 8 / 9 with
    the deepest rings the LDS holds); 10..12: 128 x 256 / 256 x 128 on four waves."""
    variants = [0, 1, 2] if dtype == OSD_BF16 else [0]
    if dtype == OSD_BF16:
        variants.append(15)         # variant 0 with the DMA pieces issued between the MFMA rows
    if dtype == OSD_BF16 and cout >= 256 and cin >= 256:
        variants += [3, 4, 5, 8, 9, 13]     # 13, 3: the software-pipelined kernel, splits / team mode (refused where they do not apply)
    if dtype == OSD_BF16 and cin >= 256:
        variants += [6, 10]         # 128 co x 256 ci
    if dtype == OSD_BF16 and cout >= 256:
        variants += [7, 11, 14]     # 256 co x 128 ci (14: 11 interleaved)
    # variant 3 (team mode) reads the code as a round count and knows 0..3
    excl = [int(v) for v in os.environ.get("OSD_WGRAD_EXCLUDE", "").split(",") if v]      # A/B timing: leave variants out
    variants = [v for v in variants if v not in excl]
    return [1 + v + 16 * t for t in (0, 1, 2, 3, 4, 5, 6, 7) for v in variants if not (v == 3 and t > 3)]


def _candidate_runs(fn):
    """A tuner candidate whose kernel does not cover this geometry (OSD_ERR_UNSUPPORTED = -2: e.g. the pipelined weight-gradient
    variant on a map whose width is not a power of two) is skipped; any other failure is an error."""
    try:
        fn()
        return True
    except _lib.OsdError as e:
        if getattr(e, "code", 0) == -2:
            return False
        raise


def wgrad_timer():
    """how weight-gradient candidates are timed: back to back (warm), or — OSD_TUNE_WGRAD_COLD=1 — each launch behind a cache flush:
    inside the step a weight gradient reads forward activations written milliseconds earlier (HBM) and gradients of the last ~ms"""
    return _time_launches_cold if os.environ.get("OSD_TUNE_WGRAD_COLD") else _time_launches


def _tune_wgrad(key, d, launch, dw, db, widths=None):
    """Time every candidate on scratch outputs (the kernel accumulates) and cache the winner for this shape."""
    _time_launches = wgrad_timer()
    sdw = torch.empty_like(dw)
    sdb = None if db is None else torch.empty_like(db)
    timed = []
    cands = wgrad_algo_candidates(d.dtype, d.cout, d.cin)
    if os.environ.get("OSD_WGRAD_XR"):      # the filter-row kernel: correct, never the winner so far — opt-in
        cands = cands + wgrad_xr_candidates(d.dtype, d.cout, d.cin, d.r, d.s, d.stride_h, d.pad_h, widths)
    verbose = os.environ.get("OSD_TUNE_VERBOSE")
    for algo in cands:
        d.algo = algo
        if not _candidate_runs(lambda: launch(sdw, sdb)):
            if verbose:
                print("   wgrad tuner: variant %d code %d refused" % ((algo - 1) & 15, (algo - 1) >> 4))
            continue
        t = _time_launches(lambda: launch(sdw, sdb))
        if verbose:
            print("   wgrad tuner: variant %d code %d  %.1f us" % ((algo - 1) & 15, (algo - 1) >> 4, t * 1e3))
        timed.append((t, algo))

    def retime(algo):
        d.algo = algo
        return _time_launches(lambda: launch(sdw, sdb))
    best = _photo_finish(timed, retime)
    WGRAD_ALGO_CACHE[key] = best
    TUNE_LOG["WGRAD_ALGO_CACHE"][key] = sorted(timed)
    return best


def refine_in_step(step, sync, n_steps=12, margin=0.10, min_share=0.004, gain=0.003, budget_s=100.0, verbose=False, cycle=1):
    """Second tuning phase, on the REAL objective.  The first phase ranks a shape's kernels by isolated timing — boost clock, whole
    chip, its own cache state; the step runs them beside six other streams at the power-limited clock with cold weights, and the
    two rankings differ often enough to matter (timing the latency-sized convs cold was worth +1.4 % of the step).
    Here, for every tuned shape that takes >= `min_share` of the step (lookups per step x isolated time) and has runners-up within
    `margin` of its winner, the step itself is the yardstick: `n_steps` steps with the current choice, with the alternative, and both
    again (interleaved: the chip's clock drifts); the alternative is kept only if BOTH pairs show the step at least `gain` faster.
    step(): one training step (enqueue only); sync(): device synchronise; cycle: steps per period of a workload that cycles through
    several geometries (n_steps is rounded up to a multiple, and the MEAN step decides instead of the median).
    Returns [(cache, key, old, new, ms_old, ms_new)]."""
    import time
    caches = {"ALGO_CACHE": ALGO_CACHE, "WGRAD_ALGO_CACHE": WGRAD_ALGO_CACHE, "SPLIT_CACHE": SPLIT_CACHE}

    n_steps = (n_steps + cycle - 1) // cycle * cycle

    def measure():
        evs = []
        for _ in range(cycle):                # absorb the switch (and stay in phase with a cycling workload)
            step()
        for _ in range(n_steps):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
            step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        sync()
        ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        return ms[len(ms) // 2] if cycle == 1 else sum(ms) / len(ms)

    for c in caches.values():
        c.hits = {}
        c.census = True
    try:
        step()
        sync()
    finally:
        for c in caches.values():
            c.census = False
    step_ms = measure()
    todo = []
    for name, cache in caches.items():
        for key, hits in cache.hits.items():
            log = TUNE_LOG[name].get(key)
            if not log or key not in cache:
                continue
            cur = cache[key]
            t_cur = min([t for t, a in log if a == cur] or [log[0][0]])
            share = hits * t_cur / step_ms
            alts = [a for t, a in log if a != cur and t <= t_cur * (1.0 + margin)][:2]
            if share >= min_share and alts:
                todo.append((share, name, key, alts))
    todo.sort(key=lambda t: -t[0])
    t0, changed = time.time(), []
    for share, name, key, alts in todo:
        cache = caches[name]
        for alt in alts:
            if time.time() - t0 > budget_s:
                break
            base = cache[key]
            a1 = measure()
            cache[key] = alt
            b1 = measure()
            cache[key] = base
            a2 = measure()
            cache[key] = alt
            b2 = measure()
            keep = b1 < a1 * (1.0 - gain) and b2 < a2 * (1.0 - gain)
            if verbose:
                print("   refine %s share %.3f: %s -> %s  %.3f/%.3f  %.3f/%.3f ms  %s" % (name, share, base, alt, a1, b1, a2, b2, "KEEP" if keep else "-"), flush=True)
            if keep:
                changed.append((name, key, base, alt, 0.5 * (a1 + a2), 0.5 * (b1 + b2)))
            else:
                cache[key] = base
    return changed
