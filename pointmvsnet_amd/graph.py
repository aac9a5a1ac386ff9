"""hipGraph capture of the whole inference forward.

The eager forward of BASELINE config 2 is ~330 kernel launches; at ~5 ms of GPU work it is host-bound
(Python + launch overhead > kernel time).  ``GraphedForward`` captures ``PointMVSNet.run`` -- device-only
by construction, see ``ScenePlan`` -- once, then serves every scene of the same shape by
(1) redoing the host camera algebra into the plan's pinned block + one async H2D copy (kept OUTSIDE the graph:
    as the graph's first node, with the host waiting for the previous replay before refilling the pinned block, it
    measured 587 against 600 depth maps/s, profiles/archive/r02/r02ad_graph_input_ab.log),
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
import contextlib
import copy
import gc

import torch

from . import pointflow


@contextlib.contextmanager
def capturing(graph, **kw):
    """``torch.cuda.graph(graph)`` with the cyclic garbage collector OFF for the duration of the capture.  A collection
    that starts inside a capture may finalize another object that owns a hipGraph or device memory (a _ModuleGraph of a
    model that went out of scope, a lane of an earlier test) -- destroying a graph / freeing into the pool while a stream
    is capturing aborts the process (round 6: `Fatal Python error: Aborted ... Garbage-collecting` inside a lane's capture,
    whole-suite run on hardware).  torch.cuda.graph itself collects once BEFORE the capture starts; nothing stops a
    threshold-triggered collection DURING it."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was:
            gc.enable()


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
        self._lane, self._level = pointflow.current_lane(), pointflow.CONCURRENCY      # re-captures keep both
        buffers = [(b, b.clone()) for b in model.buffers()]          # warm-up must not advance BN statistics
        for _ in range(warmup):                            # lazy allocations, packed weights, code objects
            model.run(self.plan, self.static_img, isFlow)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, saved in buffers:
                b.copy_(saved)
        self._capture()

    def _capture(self):
        # One graph for the whole forward; the stream forks inside run() become graph edges.  (Three graphs
        # -- coarse, flow tower on a side stream, flow iterations -- ordered by stream events were measured
        # at 402 depth maps/s against 486: replays on different streams did not overlap, profiles/archive/r01/r01h_split_ab.log.)
        pointflow.pack_unpin(self._packs)
        self.graph = torch.cuda.CUDAGraph()
        lane_now = pointflow.current_lane()
        pointflow.set_lane(self._lane)
        pointflow.pack_log_begin()
        try:
            with pointflow.concurrency(self._level), capturing(self.graph):
                self.outputs = self.model.run(self.plan, self.static_img, self.isFlow)
        finally:
            self._packs = pointflow.pack_log_end(pin=True)
            pointflow.set_lane(lane_now)
        self._watched = [t for t in list(self.model.parameters()) + list(self.model.buffers())]
        self._addresses = [t.data_ptr() for t in self._watched]
        self._modes = [m.training for m in self.model.modules()]      # train()/eval() picks other kernels: re-capture
        self._replays = 0

    def _rebound(self):
        """A parameter / buffer was re-bound to other storage since capture (the graph holds its old address).
        ``param.data = ...`` / a swapped buffer keep the tensor OBJECT and move its storage, so the addresses of the
        objects seen at capture are what is compared on every replay (one data_ptr() per tensor: ~0.03 ms for the
        ~340 tensors; the module tree is walked again only every 64th replay -- a registered / removed module, or a
        parameter replaced by ``setattr``, changes the objects)."""
        self._replays += 1
        if self._replays % 64 == 0:
            now = list(self.model.parameters()) + list(self.model.buffers())
            if len(now) != len(self._watched) or any(a is not b for a, b in zip(now, self._watched)):
                return True
        return any(t.data_ptr() != a for t, a in zip(self._watched, self._addresses))

    def invalidate(self):
        """Re-capture on the next call.  The storage address of every parameter / buffer seen at capture is checked on
        EVERY replay, the module tree and the sub-modules' training flags only every 64th (a walk of ~340 objects costs
        more than the replay's own enqueue): after ``setattr(module, name, new_parameter)``, registering / removing a
        sub-module, or ``child.eval()`` on a sub-module, call this instead of waiting out that window."""
        self._modes = None

    def __del__(self):
        try:
            pointflow.pack_unpin(self._packs)
        except Exception:
            pass

    def __call__(self, data_batch):
        if (self._modes is None                                            # invalidate()
                or pointflow.pack_entries_stale(self._packs) or self._rebound()   # the graph holds old packs / addresses
                or self.model.training != self._modes[0]                   # net.train() / net.eval(): every replay
                or (self._replays % 64 == 0                                # a sub-module's own flag: every 64th
                    and self._modes != [m.training for m in self.model.modules()])):
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
    intermediates and BatchNorm buffers (``replicate_for_lane``), replayed round-robin on ``lanes`` streams.

    Why: a depth map is a chain of ~85 dependent kernels, most of them too small to fill 256 CUs and each paying ~5 us
    of dependency latency; other scenes fill those holes.  Measured on MI355X (BASELINE config 2,
    profiles/archive/r03/r03b_lanes_queues.md): 640 depth maps/s with one lane, 895 with two, 970 with three, 1030 with four
    (more lanes add nothing: four hardware queues).

    Every lane is captured as a single chain (``pointflow.concurrency(0)``): forks inside the graphs take hardware
    queues away from the lanes (3 lanes: 771 / 938 / 808 depth maps/s at intra-forward levels 0 / 1 / 2 -- erratic,
    because which streams share a queue changes with every fork).  WHICH streams the lanes run on matters as much:
    HIP maps streams to its hardware queues (GPU_MAX_HW_QUEUES, default 4) by rules that depend on the process's
    history -- four streams created first thing in the process gave 860 depth maps/s, four created after the capture
    1030, with the same graphs; two queues congruent modulo 4 (GPU_MAX_HW_QUEUES = 8) gave 530, less than one lane.
    A graph can be replayed on any stream, so the placement is MEASURED: ``calibrate`` candidate stream sets each
    replay a short burst of the captured graphs (BatchNorm buffers restored afterwards) and the fastest set is kept
    (``placement`` holds the rates).

    Depth maps are bit-identical to the single-lane forward's: nothing is shared between lanes but read-only weights.
    ``submit(batch)`` enqueues one scene on the next lane and returns ``(lane, outputs)``; the outputs are that lane's
    static tensors -- valid once ``streams[lane]`` has been waited for (``wait(lane)``), overwritten by the lane's
    next ``submit``."""

    def __init__(self, model, example_batch, img_scales, inter_scales, isFlow=True, isTest=True, lanes=4, warmup=3,
                 concurrency=None, streams=None, calibrate=3):
        if lanes < 1:
            raise ValueError("LanedForward: lanes >= 1")
        self.lanes = int(lanes)
        dev = example_batch["img_list"].device
        self.models = [model] + [replicate_for_lane(model) for _ in range(self.lanes - 1)]
        level = (0 if self.lanes > 1 else pointflow.CONCURRENCY) if concurrency is None else int(concurrency)
        self.graphs = []
        try:
            with pointflow.concurrency(level):
                for lane in range(self.lanes):
                    pointflow.set_lane(lane)
                    self.graphs.append(GraphedForward(self.models[lane], example_batch, img_scales, inter_scales,
                                                      isFlow=isFlow, isTest=isTest, warmup=warmup))
        finally:
            pointflow.set_lane(0)
        torch.cuda.synchronize(dev)
        self.placement = None
        if streams is not None:
            self.streams = list(streams[:self.lanes])
        else:
            candidates = [[torch.cuda.Stream(device=dev) for _ in range(self.lanes)]
                          for _ in range(max(1, int(calibrate)) if self.lanes > 1 else 1)]
            self.streams = candidates[0]
            if len(candidates) > 1:
                self.placement = [self._burst(c) for c in candidates]
                self.streams = candidates[max(range(len(candidates)), key=lambda i: self.placement[i])]
        self._next = 0

    def _burst(self, streams, scenes_per_lane=6):
        """Scenes per second of a short burst of replays on ``streams`` (the static inputs as they are); the modules'
        buffers come back unchanged."""
        import time
        saved = [[(b, b.clone()) for b in m.buffers()] for m in self.models]
        rate = 0.0
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(self.lanes * scenes_per_lane):
                with torch.cuda.stream(streams[i % self.lanes]):
                    self.graphs[i % self.lanes].graph.replay()
            torch.cuda.synchronize()
            rate = self.lanes * scenes_per_lane / (time.perf_counter() - t0)
        with torch.no_grad():
            for per_model in saved:
                for b, old in per_model:
                    b.copy_(old)
        torch.cuda.synchronize()
        return rate

    def _sync_replica(self, lane):
        """Lane replicas own their BatchNorm buffers and ``training`` flags (replicate_for_lane): ``net.eval()`` /
        ``net.train()`` on the master, or ``load_state_dict`` (which copies into the MASTER'S buffers only), would
        otherwise never reach lanes >= 1.  The modes are compared on every submit (cheap); when they differ the replica
        takes the master's modes AND buffers (its GraphedForward then re-captures itself: it watches the modes).  The
        root module's flag is what is compared -- the host paces four lanes at ~1 000 scenes/s, a walk over the ~200
        sub-modules per submit would cost a few per cent of that.
        ``sync_buffers()`` copies the buffers on request after a ``load_state_dict``."""
        master, rep = self.models[0], self.models[lane]
        if lane == 0 or master.training == rep.training:       # (net.train() / net.eval() set every sub-module)
            return False
        for a, b in zip(master.modules(), rep.modules()):
            b.training = a.training
        torch.cuda.synchronize()
        self._copy_buffers(lane)
        return True

    def _copy_buffers(self, lane):
        with torch.no_grad():
            for a, b in zip(self.models[0].buffers(), self.models[lane].buffers()):
                b.copy_(a)

    def invalidate(self):
        """Every lane re-captures on its next submit, after taking the master's sub-module modes and buffers: call after
        structural changes of the master (a Parameter replaced by ``setattr``, a sub-module's own ``eval()``), which
        the per-submit checks (storage addresses, the ROOT module's training flag) do not see -- see
        GraphedForward.invalidate."""
        torch.cuda.synchronize()
        master = self.models[0]
        for lane in range(self.lanes):
            if lane:
                rep = self.models[lane]
                for (_, a), (_, b) in zip(master.named_modules(), rep.named_modules()):
                    b.training = a.training
                    for name, p in a._parameters.items():
                        b._parameters[name] = p
                self._copy_buffers(lane)
            self.graphs[lane].invalidate()

    def sync_buffers(self):
        """Copy the master's buffers (BatchNorm running statistics) into every lane replica -- call after
        ``load_state_dict`` on the model when eval-mode BatchNorm will read them."""
        torch.cuda.synchronize()
        for lane in range(1, self.lanes):
            self._copy_buffers(lane)

    def submit(self, data_batch):
        lane = self._next
        self._next = (lane + 1) % self.lanes
        st = self.streams[lane]
        # the batch was produced on the caller's stream (e.g. ``v.cuda(non_blocking=True)`` in a loader loop): the lane
        # must not read it before that work is done, and the caching allocator must not recycle its memory while the
        # lane still copies from it
        st.wait_stream(torch.cuda.current_stream())
        for v in data_batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(st)
        self._sync_replica(lane)
        with torch.cuda.stream(st):
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


# ---------------------------------------------------------------------------------------------
# Module-level replay for the DROP-IN route (round 6)
# ---------------------------------------------------------------------------------------------
# The reference's model.py calls the operator modules one at a time (model.py:71-77, 113, 140-148, 211-219): 6 ImageConv
# forwards, 1 VolumeConv, 15 EdgeConv and 5 SharedMLP forwards per depth map at cfg 2 -- ~230 kernel launches of this
# package issued from Python between the ATen calls of model.py, on a route that model.py's own host synchronisations
# (linspace / view on device scalars, torch.tensor(...).to(device), torch.inverse) keep host-bound
# (profiles/r06b_route_profile.md: 12.8 ms per depth map, the GPU idle most of it).  A module's inference forward is
# device-only and a pure function of (inputs, parameters, buffers), so it is captured ONCE per input signature in a
# hipGraph and replayed: one launch instead of 10-35.  Switched on by compat.install_as_pointmvsnet() (MODULE_GRAPHS);
# the plain package default is off (module forwards launch their kernels eagerly).
MODULE_GRAPHS = False
MODULE_GRAPH_ENTRIES = 6             # signatures kept per module (least recently used goes first)


class _ModuleGraph(object):
    """One captured inference forward of ``module`` for one input signature."""

    def __init__(self, module, fn, tensors):
        self.module = module
        self.static_in = [t.detach().clone(memory_format=torch.contiguous_format) for t in tensors]
        self.graph = torch.cuda.CUDAGraph()
        pointflow.pack_log_begin()
        try:
            with torch.no_grad(), capturing(self.graph):
                self.out = fn(*self.static_in)
        finally:
            self.packs = pointflow.pack_log_end(pin=True)
        self.watched = list(module.parameters()) + list(module.buffers())
        self.addresses = [t.data_ptr() for t in self.watched]
        self.mods = list(module.modules())
        self.bns = [m for m in self.mods if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        self.state = self._state()
        self.replays = 0

    def _state(self):
        """What a captured forward bakes in besides addresses: every sub-module's train / eval flag and the BatchNorm
        hyper-parameters that travel as kernel arguments (momentum, eps, track_running_stats)."""
        return ([m.training for m in self.mods],
                [(b.momentum, b.eps, b.track_running_stats, b.running_mean is None) for b in self.bns])

    def stale(self):
        """Checked on EVERY replay: the packed weights' sources (version counter, storage), the storage address of every
        parameter / buffer seen at capture, the sub-modules' modes and BatchNorm hyper-parameters.  The module TREE is
        walked again only every 64th replay (a registered / removed sub-module or a parameter replaced by setattr
        changes the objects): the walk costs more than the replay's own enqueue."""
        self.replays += 1
        if self.replays % 64 == 0:
            if (len(self.mods) != sum(1 for _ in self.module.modules())
                    or any(a is not b for a, b in zip(self.watched,
                                                      list(self.module.parameters()) + list(self.module.buffers())))):
                return True
        return (pointflow.pack_entries_stale(self.packs)
                or any(t.data_ptr() != a for t, a in zip(self.watched, self.addresses))
                or self.state != self._state())

    def release(self):
        pointflow.pack_unpin(self.packs)
        self.packs = []

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def replay(self, tensors):
        for s, t in zip(self.static_in, tensors):
            s.copy_(t, non_blocking=True)
        self.graph.replay()
        return _clone_tree(self.out)         # the static outputs are overwritten by the next replay


def _clone_tree(x):
    if torch.is_tensor(x):
        return x.clone()
    if isinstance(x, dict):
        return type(x)((k, _clone_tree(v)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return type(x)(_clone_tree(v) for v in x)
    return x


def module_forward(module, fn, *tensors):
    """``fn(*tensors)`` -- the device-only inference forward of ``module`` -- through the module's graph cache when
    MODULE_GRAPHS is on: the FIRST call with a signature runs eagerly (lazy allocations, packed weights; it is an
    ordinary forward), the second captures and replays, later ones replay.  BatchNorm running statistics and
    ``num_batches_tracked`` advance on every replay exactly as eagerly (the update kernels are nodes of the graph).  A
    replay after an optimizer step / ``load_state_dict`` / ``.to()`` / ``train()`` / ``eval()`` / a changed BatchNorm
    momentum re-captures instead of serving the old state (same checks as GraphedForward, over this module's tensors)."""
    if not MODULE_GRAPHS or torch.cuda.is_current_stream_capturing():
        return fn(*tensors)
    key = tuple((tuple(t.shape), t.dtype, t.device.index) for t in tensors)
    cache = module.__dict__.setdefault("_pf_graphs", {})
    entry = cache.get(key)
    if entry is None:
        cache[key] = False                   # seen once: eager now, capture next time
        return fn(*tensors)
    if entry is False or entry.stale():
        if entry is not False:
            torch.cuda.synchronize()
            entry.release()
        while len(cache) > MODULE_GRAPH_ENTRIES:
            old = next(iter(cache))
            if cache[old] is not False:
                torch.cuda.synchronize()
                cache[old].release()
            del cache[old]
        entry = _ModuleGraph(module, fn, tensors)
    else:
        del cache[key]                       # (re-inserted below: dict order = recency)
    cache[key] = entry
    return entry.replay(tensors)
