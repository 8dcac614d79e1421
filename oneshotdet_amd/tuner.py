"""Per-shape algorithm selection by measurement ("measure, don't guess"): the caches ops.py consults, the candidate lists and
the timing loops.  `with tuning(): ...` makes every conv / weight-gradient shape met for the first time time all of its
candidates on the device and cache the winner; without it an uncached shape runs the library's default (algo 0).
dist_utils.broadcast_tuner_choices / save_tuner_choices / load_tuner_choices move the three caches between ranks and runs."""
import os

import torch

from . import _lib
from ._lib import OSD_BF16

SPLIT_CACHE = {}          # grouped launches: where to cut the segment list into a large-level and a small-level launch
WGRAD_ALGO_CACHE = {}     # weight-gradient launches: 1 + variant + 16 * split-target code

# osd_conv_desc.algo = 1 + impl*32 + variant*8 + tile; impl 0 = LDS-DMA ring kernel (variants: deep / shallow ring /
# short stages), impl 1 = register-staged kernel; tile 0..4 = 128x128, 128x64, 64x64, 256x16, 256x256/8 waves, 5 = 128x256/8 waves
# (round 5; the ping-pong 256x256 kernel of rounds 1-4 is retired), 6 = row-reuse 3x3 conv_sp (variant 0: 128 x 128 tiles, round 5 — the id of the
# retired conv_xr; 1: any width; 2: the general-width form forced; 3: 128-pixel tiles), 7 = 256x128/8 waves (pixels x channels).
ALGO_CACHE = {}
_TUNING = [False]


CONV_ALGO_PX = 1 + 1 * 32 + 2 * 8 + 0      # conv_px.hip, eight waves of 16 pixels
CONV_ALGO_PX_WIDE = CONV_ALGO_PX + 1       # ... four waves of 32 pixels


def conv_algo_candidates(cout_store, relu_in, has_mask=False):
    tiles = [3] if cout_store <= 16 else [0, 1, 2]
    if cout_store <= 16:
        tiles = [3, 2]
    cands = [1 + 0 * 32 + v * 8 + t for v in (0, 1, 2, 3) for t in tiles]      # v 3: three short stages (many workgroups per CU)
    if cout_store >= 256 and not relu_in:
        cands.append(1 + 0 * 32 + 1 * 8 + 4)          # 256x256 tile, shallow ring
        cands.append(1 + 0 * 32 + 2 * 8 + 4)          # 256x256 tile, short stages x 4
        if not os.environ.get("OSD_NO_SP"):           # (A/B switch for tools and benches)
            cands.append(1 + 0 * 32 + 1 * 8 + 6)      # 3x3/1: pixel rows fetched once per filter row, software-pipelined fragments + mid-stage barrier, ANY width (bf16)
            cands.append(1 + 0 * 32 + 3 * 8 + 6)      # ... on 128-pixel tiles: twice the workgroups where 256-pixel tiles leave CUs idle
            cands.append(1 + 0 * 32 + 0 * 8 + 6)      # ... on 128 x 128 tiles: half the weight stream per workgroup (layer4's 3x3: K = 4,608 on 6,400 pixels)
    if 64 < cout_store <= 128 and not relu_in and not os.environ.get("OSD_NO_SP") and not os.environ.get("OSD_NO_SP_NARROW"):
        cands.append(1 + 0 * 32 + 1 * 8 + 6)          # the same kernel on a 256-pixel x 128-channel tile (4 x 2 waves): layer2's 3x3 convs
    if cout_store >= 128 and not relu_in:
        cands += [1 + 0 * 32 + v * 8 + 7 for v in (0, 1, 2, 3)]      # 256x128 tile on 8 waves: deep / shallow ring / short stages
    if cout_store >= 256 and not relu_in:
        cands += [1 + 0 * 32 + v * 8 + 5 for v in (0, 1, 2, 3)]      # 128x256 tile on 8 waves: all of N = 256 per pixel tile (reducing 1x1 convs)
    if not relu_in and not has_mask:
        cands += [1 + 1 * 32 + t for t in tiles]
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


def _tune(key, d, launch, cands=None):
    """Time every candidate algorithm for this conv shape (launch() reads d.algo) and cache the fastest."""
    timed = []
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
    for algo in (cands if cands is not None else conv_algo_candidates(d.cout, d.relu_in, has_mask=bool(key[-1]))):
        d.algo = algo
        try:
            launch()
        except _lib.OsdError:
            continue
        timed.append((_time_launches(launch), algo))

    def retime(algo):
        d.algo = algo
        return _time_launches(launch)
    best = _photo_finish(timed, retime)
    ALGO_CACHE[key] = best
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


def _tune_wgrad(key, d, launch, dw, db, widths=None):
    """Time every candidate on scratch outputs (the kernel accumulates) and cache the winner for this shape."""
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
    return best
