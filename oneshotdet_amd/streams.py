"""Cheap stream plumbing for the host side of a step (one process drives one GPU).

`torch.cuda.current_stream()` and `with torch.cuda.stream(s)` go through `_lazy_init`, `_get_device_index` and `is_available`
(an `os.environ` lookup) on every call: ~8 us each, ~500 times per training step — a third of the host's enqueue time, and the
multi-scale workload of BASELINE.json configs[4] is HOST-bound (tools/config5_host_probe.py).  These helpers call the
C entry points those functions end in.  Same semantics as the torch forms for streams of the CURRENT device; anything else (or a
torch build without the private entry points) takes the torch forms."""
import torch

_get_cur = getattr(torch._C, "_cuda_getCurrentStream", None)
_set = getattr(torch._C, "_cuda_setStream", None)
_get_raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_dev = getattr(torch._C, "_cuda_getDevice", None)
FAST = all(f is not None for f in (_get_cur, _set, _get_raw, _get_dev))


def current():
    """torch.cuda.current_stream() of the current device"""
    if FAST:
        try:
            sid, idx, typ = _get_cur(_get_dev())
            return torch.cuda.Stream(stream_id=sid, device_index=idx, device_type=typ)
        except Exception:
            pass
    return torch.cuda.current_stream()


def raw_handle():
    """the current stream's hipStream_t as an int"""
    if FAST:
        try:
            return _get_raw(_get_dev())
        except Exception:      # context not initialised yet: the torch form initialises it
            pass
    return torch.cuda.current_stream().cuda_stream


class on(object):
    """`with streams.on(s):` = `with torch.cuda.stream(s):` for a stream of the current device (None: no-op)"""
    __slots__ = ("s", "prev", "slow")

    def __init__(self, s):
        self.s, self.prev, self.slow = s, None, None

    def __enter__(self):
        s = self.s
        if s is None:
            return
        if FAST:
            try:
                dev = _get_dev()
                if s.device_index == dev:
                    self.prev = _get_cur(dev)
                    _set(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
                    return
            except Exception:
                self.prev = None
        self.slow = torch.cuda.stream(s)
        self.slow.__enter__()

    def __exit__(self, *a):
        if self.slow is not None:
            slow, self.slow = self.slow, None
            return slow.__exit__(*a)
        p, self.prev = self.prev, None
        if p is not None:
            _set(stream_id=p[0], device_index=p[1], device_type=p[2])
