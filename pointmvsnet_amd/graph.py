"""hipGraph capture of the whole inference forward.

The eager forward of BASELINE config 2 is ~330 kernel launches; at ~5 ms of GPU work it is host-bound
(Python + launch overhead > kernel time).  ``GraphedForward`` captures ``PointMVSNet.run`` -- device-only
by construction, see ``ScenePlan`` -- once, then serves every scene of the same shape by
(1) redoing the host camera algebra into the plan's pinned block + one async H2D copy (kept OUTSIDE the graph:
    as the graph's first node, with the host waiting for the previous replay before refilling the pinned block, it
    measured 587 against 600 depth maps/s, profiles/r02ad_graph_input_ab.log),
(2) copying the images into the static input buffer -- skipped when the caller's images already live there:
    ``adopt_input=True`` makes the example batch's own image tensor the static input (a data loader that fills a
    ring of input buffers keeps one GraphedForward per slot and never copies),
(3) one ``hipGraphLaunch``.
BatchNorm running statistics and ``num_batches_tracked`` keep mutating on every replay exactly as in
eager mode (the update kernels are part of the graph).  Outputs are static tensors, overwritten by the
next replay.

Weights: the graph reads the PACKED copies of the conv / GEMM weights (pointflow's pack cache).  The entries
used during capture are pinned (never evicted while this object lives) and their source parameters are
fingerprinted (version counter, storage address, device, dtype); a replay after ``load_state_dict``, an
optimizer step, ``param.data = ...`` or ``model.to(...)`` re-captures instead of silently serving the old
weights.  Everything else the kernels read through RAW pointers baked into the graph (BatchNorm gamma / beta / running
statistics inside the pf_bn_job blocks, the flow head's weight) is watched by storage address: rebinding one of those
tensors (``bn.weight.data = ...``, a swapped buffer) re-captures too; in-place updates need nothing, the pointer sees
them.  The warm-up forwards run on a snapshot of the BatchNorm buffers, which is restored before capture:
constructing a GraphedForward does not advance running statistics or ``num_batches_tracked``.
"""
import copy

import torch

from . import pointflow


class GraphedForward(object):
    def __init__(self, model, example_batch, img_scales, inter_scales, isFlow=True, isTest=True, warmup=3,
                 adopt_input=False):
        if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
            raise RuntimeError("GraphedForward captures the inference path; wrap the call in torch.no_grad()")
        self.model = model
        self.isFlow = isFlow
        self.probe = None                                  # set to a list to collect (start, end) events per replay
        self.static_img = example_batch["img_list"] if adopt_input else example_batch["img_list"].clone()
        self.plan = model.make_plan(example_batch, img_scales, inter_scales, isTest)
        self._packs = []
        self.recaptures = 0
        buffers = [(b, b.clone()) for b in model.buffers()]          # warm-up must not advance BN statistics
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # library warm-up (MIOpen find, lazy allocations)
            for _ in range(warmup):
                model.run(self.plan, self.static_img, isFlow)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, saved in buffers:
                b.copy_(saved)
        self._capture()

    def _capture(self):
        # One graph for the whole forward; the stream forks inside run() become graph edges.  (Three graphs
        # -- coarse, flow tower on a side stream, flow iterations -- ordered by stream events were measured
        # at 402 depth maps/s against 486: replays on different streams did not overlap, profiles/r01h_split_ab.log.)
        pointflow.pack_unpin(self._packs)
        self.graph = torch.cuda.CUDAGraph()
        pointflow.pack_log_begin()
        try:
            with torch.cuda.graph(self.graph):
                self.outputs = self.model.run(self.plan, self.static_img, self.isFlow)
        finally:
            self._packs = pointflow.pack_log_end(pin=True)
        self._watched = [t for t in list(self.model.parameters()) + list(self.model.buffers())]
        self._addresses = [t.data_ptr() for t in self._watched]

    def _rebound(self):
        """A parameter / buffer was re-bound to other storage since capture (the graph holds its old address)."""
        now = list(self.model.parameters()) + list(self.model.buffers())
        return len(now) != len(self._watched) or [t.data_ptr() for t in now] != self._addresses

    def __del__(self):
        try:
            pointflow.pack_unpin(self._packs)
        except Exception:
            pass

    def __call__(self, data_batch):
        if pointflow.pack_entries_stale(self._packs) or self._rebound():   # the graph holds old packs / addresses
            torch.cuda.synchronize()
            self._capture()
            self.recaptures += 1
        self.plan.update_(data_batch)
        img = data_batch["img_list"]
        if img.data_ptr() != self.static_img.data_ptr():
            self.static_img.copy_(img, non_blocking=True)
        if self.probe is not None:                         # diagnostic: GPU-side extent of each replay
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.graph.replay()
            b.record()
            self.probe.append((a, b))
        else:
            self.graph.replay()
        return self.outputs


def replicate_for_lane(model):
    """A replica of ``model`` that SHARES its parameters (the very same Parameter objects: an optimizer step or a
    ``load_state_dict`` on either is seen by both) and OWNS its buffers -- ``nn.DataParallel``'s replica semantics
    (reference test.py:84, train.py:177) on one device.  Every scene lane advances its own BatchNorm running statistics
    and ``num_batches_tracked`` (in the order of the scenes it sees; batch statistics, hence depth maps, never read
    them in the reference's train()-mode evaluation, test.py:58); lane 0 is the module itself."""
    held = {k: model.__dict__[k] for k in ("_plan", "_tplan") if k in model.__dict__}
    for k in held:                                  # per-call caches (pinned blocks, events) are not part of the module
        model.__dict__[k] = None
    try:
        replica = copy.deepcopy(model)
    finally:
        model.__dict__.update(held)
    for (_, master), (_, rep) in zip(model.named_modules(), replica.named_modules()):
        for name, p in master._parameters.items():
            rep._parameters[name] = p
    return replica


class LanedForward(object):
    """Several scenes in flight on ONE GPU: ``lanes`` captured forwards, each with its own static input, plan block,
    intermediates, auxiliary streams and BatchNorm buffers (``replicate_for_lane``), replayed round-robin on ``lanes``
    streams.  A depth map is a chain of ~85 dependent kernels, most of them too small to fill 256 CUs and each paying
    ~5 us of dependency latency; a second scene fills those holes (measured on MI355X, BASELINE config 2: 669 -> 745
    depth maps/s with 2 lanes and the host still in the way, profiles/r03_lanes_ab.log).  Depth maps are bit-identical
    to the single-lane forward's: nothing is shared between lanes but read-only weights.

    ``submit(batch)`` enqueues one scene on the next lane and returns ``(lane, outputs)``; the outputs are that lane's
    static tensors -- valid once ``streams[lane]`` has been waited for (``wait(lane)``), overwritten by the lane's
    next ``submit``."""

    def __init__(self, model, example_batch, img_scales, inter_scales, isFlow=True, isTest=True, lanes=2, warmup=3):
        if lanes < 1:
            raise ValueError("LanedForward: lanes >= 1")
        self.lanes = int(lanes)
        self.models = [model] + [replicate_for_lane(model) for _ in range(self.lanes - 1)]
        self.streams = [torch.cuda.Stream(device=example_batch["img_list"].device) for _ in range(self.lanes)]
        self.graphs = []
        here = torch.cuda.current_stream()
        try:
            for lane in range(self.lanes):
                pointflow.set_lane(lane)
                self.streams[lane].wait_stream(here)
                with torch.cuda.stream(self.streams[lane]):
                    self.graphs.append(GraphedForward(self.models[lane], example_batch, img_scales, inter_scales,
                                                      isFlow=isFlow, isTest=isTest, warmup=warmup))
        finally:
            pointflow.set_lane(0)
        self._next = 0

    def submit(self, data_batch):
        lane = self._next
        self._next = (lane + 1) % self.lanes
        with torch.cuda.stream(self.streams[lane]):
            out = self.graphs[lane](data_batch)
        return lane, out

    def wait(self, lane=None):
        """Make the CURRENT stream wait for ``lane`` (all lanes when None)."""
        cur = torch.cuda.current_stream()
        for i in (range(self.lanes) if lane is None else (lane,)):
            cur.wait_stream(self.streams[i])

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
