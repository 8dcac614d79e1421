"""Launch trace (tests only).  With TRACE = [] every wrapper in ops.py appends (kind, {operands and parameters}) after its
launch, holding references to the device tensors it read and wrote.  tests/test_gpu_launch_replay.py replays every recorded
launch of a real step on the CPU (oracle/launch_replay.py) from the engine's OWN inputs and compares the outputs.  Launches
made while the tuner is timing candidates are not recorded."""
from .tuner import _TUNING

TRACE = None


def rec(kind, **kw):
    if TRACE is not None and not _TUNING[0]:
        TRACE.append((kind, kw))
