"""Fork/join helper: run independent branches of the forward on side HIP streams.

Inside one encoder stage the three temporal views are independent chains of Swin blocks; view 1/2 GEMMs are small
(M = B*196 rows in stage 2) and leave most CUs idle when they run alone.  Forking them onto side streams lets the
hardware co-schedule them with view 3's large kernels; under hipGraph capture the forks become parallel graph branches.
Results are bitwise identical to the serial order (no kernel depends on launch order; no atomics anywhere).

Set MUMPY_SERIAL=1 to disable (A/B timing)."""
import os

import torch
import torch.distributed

_SIDE = {}
_DEPTH = [0]            # nesting depth of run_parallel on this (host) thread: nested forks get their own side streams
SERIAL = os.environ.get("MUMPY_SERIAL", "0") == "1"


def new_distinct_stream(device, avoid=(), priority=0):
    """A torch stream whose raw HIP handle differs from every handle in `avoid` and from every side stream of this module.
    torch hands pool streams out round-robin (32 per priority): after enough requests a "new" Stream object wraps a handle
    that is already in use, which would make a fork run on its own parent (silently serial) or two graphs share a side
    stream.  Streams drawn and rejected here are dropped again (they are pool members, nothing is destroyed)."""
    taken = {int(h) for h in avoid} | {s.cuda_stream for s in _SIDE.values()}
    # With a process group alive, stay out of the pool RCCL's own stream comes from (ProcessGroupNCCL draws a NORMAL-priority pool
    # stream): a capture stream must never be a stream a pending collective is tied to -- the watchdog thread's poll of that
    # collective's end event then fails with hipErrorCapturedEvent and aborts the process (see quiesce_collectives below and
    # tools/rccl_capture_probe.py).  All of this module's streams then come from the high-priority pool (equal among themselves,
    # so the fork/join schedule is unchanged).
    if priority == 0 and torch.distributed.is_available() and torch.distributed.is_initialized():
        priority = -1
    for _ in range(64):
        s = torch.cuda.Stream(device=device, priority=priority)
        if s.cuda_stream not in taken:
            return s
    raise RuntimeError("mumpy_hip.streams: torch's stream pool is exhausted (more than 32 concurrent side streams)")


def quiesce_collectives():
    """Call before a hipGraph capture: device idle, and -- with an RCCL process group alive -- its watchdog thread given time to
    retire every finished collective (it sweeps its list every 100 ms).  A collective still on that list while a capture is open
    is polled through hipEventQuery from the watchdog thread, which on ROCm 7.2 can fail with hipErrorCapturedEvent and abort the
    process (tools/rccl_capture_probe.py); an empty list cannot."""
    torch.cuda.synchronize()
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl":
        import time
        time.sleep(0.5)


def _side_stream(device, i, parent=None):
    # keyed by the PARENT stream as well: two forks at the same depth under different parents never share a side stream
    key = (str(device), _DEPTH[0], i, None if parent is None else parent.cuda_stream)
    if key not in _SIDE:
        prio = int(os.environ.get("MUMPY_SIDE_PRIORITY", "0"))      # (high priority measured slightly slower)
        _SIDE[key] = new_distinct_stream(device, () if parent is None else (parent.cuda_stream,), prio)
    return _SIDE[key]


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)


def run_parallel(branches, inputs):
    """branches: callables; inputs: per-branch tuple of tensors the branch reads.  The LAST branch runs on the current
    stream (give it the heaviest chain); the others on side streams forked from / joined to it."""
    if SERIAL or len(branches) == 1:
        return [fn() for fn in branches]
    main = torch.cuda.current_stream()
    # Forks are rooted on the stream the caller entered with, never on one of this module's side streams: a fork from a
    # side stream inside hipGraph capture segfaulted in CUDAGraph.capture_end (ROCm 7.2; gpurun_out/crash.log of round 1:
    # the nested fork's streams joined their side-stream parent, which itself joined the capturing stream only later).
    # A nested fork reached on a side stream therefore runs its branches in order on that stream -- same kernels, same
    # results; the rule is enforced here instead of by the order in which callers list their branches.
    if any(main.cuda_stream == s.cuda_stream for s in _SIDE.values()):
        return [fn() for fn in branches]
    fork = torch.cuda.Event()
    fork.record(main)
    outs = [None] * len(branches)
    sides = []
    for i in range(len(branches) - 1):
        sides.append(_side_stream(main.device, i, main))     # streams of THIS depth and parent
    _DEPTH[0] += 1
    try:
        for i, fn in enumerate(branches[:-1]):
            s = sides[i]
            s.wait_event(fork)
            for t in _tensors(inputs[i]):
                t.record_stream(s)              # allocated on `main`, read on `s`
            with torch.cuda.stream(s):
                outs[i] = fn()
        outs[-1] = branches[-1]()
    finally:
        _DEPTH[0] -= 1
    for i, s in enumerate(sides):
        main.wait_stream(s)
        for t in _tensors(outs[i]):
            t.record_stream(main)           # allocated on `s`, consumed on `main`
    return outs
