"""Data-parallel gradient exchange: the ONE collective of the training step (tools/train_net.py:83-88 uses
DistributedDataParallel; here the gradients already live in one flat fp32 buffer, so the exchange is a few large
contiguous all-reduces — RCCL over xGMI with backend "nccl", gloo in the CPU tests)."""
import os

import torch

from . import streams
import torch.distributed as dist


def average_flat_(flat, group=None, n_buckets=4):
    """In-place average of a flat 1-D tensor over the ranks of `group`; no-op without an initialised process group or
    with a single rank.  Buckets are contiguous slices, so no packing copies are needed."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    n = flat.numel()
    if n == 0:
        return flat
    chunk = max(1, (n + n_buckets - 1) // n_buckets)
    works = [dist.all_reduce(flat[i:i + chunk], group=group, async_op=True) for i in range(0, n, chunk)]
    for w in works:
        w.wait()
    flat.mul_(1.0 / world)
    return flat


def bucket_ranges(plan, total):
    """Contiguous ranges of the flat gradient buffer in the order their gradients become final during backward.
    plan: [(parameter name, shape)] in buffer order (TrainEngine._plan); total: padded length of the buffer.
    Returns [(bucket name, lo, hi)] covering [0, total) exactly once: per backbone `layer2`, `layer3`, `layer4+fpn`
    (the FPN and layer4 finish first, layer2 last), then the FCOS head (finishes before either backbone starts) and, when
    the second stage trains too, its box head."""
    import math
    out, off = [], 0
    cur, lo = None, 0
    for name, shape in plan:
        if name.startswith("rpn."):
            b = "head"
        elif name.startswith("roi_heads."):
            b = "box_head"        # second stage (TrainEngine(second_stage=True)): final after the first-stage head, before the backbones
        else:
            bb = name.split(".", 1)[0]
            rest = name.split(".body.", 1)[1] if ".body." in name else "fpn"
            stage = rest.split(".", 1)[0]
            b = "%s.%s" % (bb, "layer4+fpn" if stage in ("layer4", "fpn") else stage)
        if b != cur:
            if cur is not None:
                out.append((cur, lo, off))
            cur, lo = b, off
        off += int(math.prod(shape))
    if cur is not None:
        out.append((cur, lo, total))
    names = [n for n, _, _ in out]
    assert len(set(names)) == len(names), "parameters of one bucket are not contiguous in the flat buffer: %s" % names
    return out


TUNER_CACHES = ("ALGO_CACHE", "SPLIT_CACHE", "WGRAD_ALGO_CACHE")


def broadcast_tuner_choices(ops, src=0, group=None):
    """Make every rank run the kernels rank `src` measured to be fastest.  The autotuner (ops.tuning()) decides by timing on the
    local GPU; with N ranks deciding independently, timing noise gives ranks different kernels for the same shape and the
    slowest choice sets the pace of every step (the step ends in a collective).  Here `src`'s caches — conv algorithm per
    shape, large / small segment cut of the grouped launches, weight-gradient variant — replace every other rank's: shapes met
    afterwards hit the cache and are never timed on those ranks.  Also usable with an env-named cache file: see
    save_tuner_choices / load_tuner_choices.  No-op without a process group or with one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    rank = dist.get_rank(group)
    payload = [{name: dict(getattr(ops, name)) for name in TUNER_CACHES} if rank == src else None]
    dist.broadcast_object_list(payload, src=src, group=group)
    if rank != src:
        for name in TUNER_CACHES:
            cache = getattr(ops, name)
            cache.clear()
            cache.update(payload[0][name])


def tuner_choices_agree(ops, group=None):
    """True iff every rank holds the same tuner caches (a digest of the sorted entries is all-gathered)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    import hashlib
    text = repr([(name, sorted((repr(k), v) for k, v in getattr(ops, name).items())) for name in TUNER_CACHES])
    mine = hashlib.sha256(text.encode()).hexdigest()
    every = [None] * dist.get_world_size(group)
    dist.all_gather_object(every, mine, group=group)
    return all(d == mine for d in every)


def _plain_key(k):
    """A tuner cache key is a (nested) tuple of ints / bools / strings / None; anything else does not belong in the file."""
    if isinstance(k, tuple):
        return tuple(_plain_key(e) for e in k)
    if k is None or isinstance(k, (bool, int, str)):
        return k
    if isinstance(k, float) and k == int(k):
        return int(k)
    raise ValueError("tuner cache key holds %r" % (k,))


def save_tuner_choices(ops, path):
    """The tuner caches as a JSON text file: {cache name: [[repr(key), int choice], ...]}; tune once, start every later run / rank
    from it (load_tuner_choices).  Plain data only — the caches map tuples of ints to ints, so nothing needs pickle (a pickle
    read back from a predictable path would run whatever a planted file holds)."""
    import json
    from . import _lib
    data = {name: sorted([repr(_plain_key(k)), int(v)] for k, v in getattr(ops, name).items()) for name in TUNER_CACHES}
    # algorithm ids are only meaningful for one library generation (ids were reused for other kernels in round 5: ADVICE r5)
    data["abi"] = int(_lib.ABI_VERSION)
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(data, f)
    os.replace(tmp, path)


def load_tuner_choices(ops, path):
    """Read a file written by save_tuner_choices: keys through ast.literal_eval (literals only), every value checked to be an int."""
    import ast
    import json
    with open(path, "r") as f:
        data = json.load(f)
    if not isinstance(data, dict):
        raise ValueError("%s: not a tuner cache file" % path)
    from . import _lib
    if data.get("abi") != _lib.ABI_VERSION:
        # written by another library generation (or before the stamp existed): its ids may name other kernels now -> tune afresh
        import warnings
        warnings.warn("%s: tuner cache of ABI %r ignored (library ABI %d)" % (path, data.get("abi"), _lib.ABI_VERSION))
        return False
    for name in TUNER_CACHES:
        for entry in data.get(name, []):
            if not (isinstance(entry, list) and len(entry) == 2 and isinstance(entry[0], str) and isinstance(entry[1], int)
                    and not isinstance(entry[1], bool)):
                raise ValueError("%s: bad entry %r in %s" % (path, entry, name))
            getattr(ops, name)[_plain_key(ast.literal_eval(entry[0]))] = entry[1]
    return True


class GradExchange(object):
    """Gradient averaging overlapped with backward: each bucket of the flat buffer is all-reduced on a communication
    stream as soon as the streams that produce it have been told everything that writes it (ready()), while the rest of
    the backward pass keeps running; finish() makes the current stream wait for all of them.  With one rank (or no
    process group) every call is a no-op.  CPU tensors (gloo tests) take the same path without streams."""

    def __init__(self, flat, ranges, group=None, single_rank_too=False, wire_dtype=None, comm_stream=None):
        self.flat, self.group = flat, group
        self.ranges = {n: (lo, hi) for n, lo, hi in ranges}
        # wire_dtype=torch.bfloat16 (device buffers only): every bucket travels as bf16 — cast on the communication stream,
        # averaged, cast back into the fp32 buffer: half the bytes per link (118 instead of 236 MB per step), one rounding of
        # every rank's contribution and one of the average; None = the reference's fp32 exchange
        self.wire_dtype = wire_dtype if (wire_dtype is not None and flat.is_cuda) else None
        self._wire = None
        import os
        # timing diagnostic for the one-rank live-exchange bench ONLY (events and streams, no collective): the subclass / tool
        # sets it; it is refused with more than one rank, where skipping the all-reduce would train on un-averaged gradients
        self._skip_collective = False
        # single_rank_too: run the collectives even with one rank (tests exercise the stream plumbing on one GPU)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank_too)
        self.world = dist.get_world_size(group) if self.active else 1
        self.cuda = flat.is_cuda
        # comm_stream: the caller's stream to run the exchange on.  The training engine hands over its update stream: a stream
        # created here, after all the others, lands on another hardware queue (streams are dealt round-robin onto the 4
        # queues a process gets) and the step measured 12 % slower with the exchange's waits in front of the main chain
        self.comm = (comm_stream if comm_stream is not None else torch.cuda.Stream(device=flat.device)) if (self.active and self.cuda) else None
        # RCCL averages natively; gloo has no AVG, so sum and scale.  Probed once with a one-element collective (every
        # rank constructs its exchange at the same point) so an unsupported op degrades to sum + scale, not to a crash
        self.avg = False
        if self.active and dist.get_backend(group) == "nccl" and self.cuda:
            try:
                probe = torch.ones(1, device=flat.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
                self.avg = abs(float(probe.item()) - 1.0) < 1e-6
            except Exception:
                self.avg = False
        self.pending = set()
        self.begin()

    def skip_collective_for_timing(self):
        """A/B diagnostic: keep the stream waits, drop the all-reduce.  One rank only."""
        if self.active and self.world > 1:
            raise RuntimeError("skip_collective_for_timing: %d ranks would train on un-averaged gradients" % self.world)
        import warnings
        warnings.warn("GradExchange: collectives are SKIPPED (timing diagnostic)")
        self._skip_collective = True

    def begin(self):
        self.pending = set(self.ranges)

    def ready(self, name, producers=()):
        """The gradients of bucket `name` are final once the work already enqueued on `producers` (streams) is done."""
        if not self.active or name not in self.pending:
            return
        self.pending.discard(name)
        lo, hi = self.ranges[name]
        if hi <= lo:
            return
        view = self.flat[lo:hi]
        if not self.cuda:
            dist.all_reduce(view, group=self.group)
            view.mul_(1.0 / self.world)
            return
        for s in producers:
            ev = torch.cuda.Event()
            ev.record(s)
            self.comm.wait_event(ev)
        if self._skip_collective:          # skip_collective_for_timing(): events and streams, no collective
            return
        with streams.on(self.comm):
            if self.wire_dtype is not None and lo % 8 == 0 and (hi - lo) % 8 == 0:
                from . import _lib, ops
                if self._wire is None:
                    self._wire = torch.empty(self.flat.numel(), device=self.flat.device, dtype=self.wire_dtype)
                wire = self._wire[lo:hi]
                _lib.call("osd_grad_wire_cast", ops._ptr(view), ops._ptr(wire), hi - lo, 1, ops._stream())
                if self.avg:
                    dist.all_reduce(wire, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(wire, group=self.group)
                    wire.mul_(1.0 / self.world)
                _lib.call("osd_grad_wire_cast", ops._ptr(wire), ops._ptr(view), hi - lo, 0, ops._stream())
            elif self.avg:
                dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(view, group=self.group)
                view.mul_(1.0 / self.world)

    def finish(self):
        """Exchange whatever has not been announced yet, then order the current stream after the whole exchange."""
        if not self.active:
            return
        cur = [streams.current()] if self.cuda else ()
        for name in [n for n in self.ranges if n in self.pending]:
            self.ready(name, cur)
        if self.cuda:
            streams.current().wait_stream(self.comm)
        self.begin()
