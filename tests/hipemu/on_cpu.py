"""``with emulated_gpu():`` -- run this package's Python layer on CPU tensors with libpointflow_emu.so behind the C ABI.

Test infrastructure (tests/test_emulated_*.py).  Inside the context
  * every C-ABI entry point resolves to the EMULATED library (tests/hipemu/build_emu.py: the kernel sources recompiled for
    the host), which takes host pointers: the "device" is the CPU;
  * the handful of torch.cuda calls the package makes around its launches (device / stream contexts, events, stream
    waits) become no-ops on one fake stream -- the emulator runs every launch to completion before it returns, so program
    order is the only order;
  * ``tensor.is_cuda`` answers True and ``record_stream`` does nothing, so the package's "is this a GPU tensor" gates take
    the paths they take on an MI355X.
Nothing here is imported by the product; outside the context torch is untouched.
"""
import contextlib
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class FakeEvent(object):
    """Launches run to completion before they return here, so an event is the wall clock at ``record()``."""

    def __init__(self, *a, **kw):
        self.t = 0.0

    def record(self, stream=None):
        import time
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def wait(self, stream=None):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class FakeStream(object):
    cuda_stream = 0
    device = torch.device("cpu")

    def __init__(self, *a, **kw):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, event):
        pass

    def record_event(self, event=None):
        return event or FakeEvent()

    def synchronize(self):
        pass

    def __eq__(self, other):
        return isinstance(other, FakeStream)

    def __hash__(self):
        return 1


class FakeGraph(object):
    """torch.cuda.CUDAGraph for dry runs: the "capture" executes its body once, eagerly; ``replay()`` does NOTHING (the
    outputs stay what the capture computed).  Enough to walk code that builds and replays graphs -- bench.py's training
    child -- not to test what a replay computes (the graph tests stay on the hardware)."""

    def __init__(self, *a, **kw):
        pass

    def replay(self):
        pass

    def raw_cuda_graph(self):
        raise RuntimeError("tests/hipemu has no hipGraphs")


_STREAM = FakeStream()


@contextlib.contextmanager
def emulated_gpu():
    sys.path.insert(0, HERE)
    import build_emu
    from pointmvsnet_amd import _lib
    lib = ctypes.CDLL(build_emu.build())
    for name, (argtypes, restype) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)                     # (AttributeError: an entry point the emulated build lacks)
        fn.argtypes, fn.restype = argtypes, restype
    null = lambda *a, **kw: contextlib.nullcontext()      # noqa: E731
    saved = []

    def patch(obj, name, value):
        saved.append((obj, name, obj.__dict__.get(name, None), name in obj.__dict__))
        setattr(obj, name, value)

    patch(_lib, "_lib", lib)
    patch(_lib, "stream", lambda: None)
    patch(_lib, "require_gpu", lambda *t: None)
    patch(torch.cuda, "device", null)
    patch(torch.cuda, "stream", null)
    patch(torch.cuda, "current_stream", lambda device=None: _STREAM)
    patch(torch.cuda, "Stream", FakeStream)
    patch(torch.cuda, "Event", FakeEvent)
    patch(torch.cuda, "synchronize", lambda device=None: None)
    patch(torch.cuda, "current_device", lambda: "cpu")
    patch(torch.cuda, "CUDAGraph", FakeGraph)
    patch(torch.cuda, "graph", null)
    patch(torch.Tensor, "is_cuda", property(lambda self: True))
    patch(torch.Tensor, "record_stream", lambda self, stream: None)
    # host == device here, so ``x.to(dev)`` would ALIAS x where on a GPU it copies: in-place kernels would then overwrite the
    # test's own reference input.  A ``.to()`` that names a device and would return the tensor itself returns a copy.
    real_to = torch.Tensor.to

    def to(self, *args, **kw):
        out = real_to(self, *args, **kw)
        names_device = "device" in kw or any(isinstance(a, (torch.device, str)) for a in args)
        if out is self and names_device:
            out = self.detach().clone().requires_grad_(self.requires_grad) if self.is_leaf else self.clone()
        return out

    patch(torch.Tensor, "to", to)
    try:
        yield lib
    finally:
        for obj, name, old, had in reversed(saved):
            if had:
                setattr(obj, name, old)
            else:
                delattr(obj, name)
