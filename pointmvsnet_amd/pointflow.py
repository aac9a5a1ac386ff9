"""Host-side orchestration of the fused PointFlow kernels.

Everything here is plumbing over the C ABI (include/pointflow_hip.h): buffer allocation with torch,
weight packing, and the launch sequence.  No arithmetic of the hot path happens in torch.

Point organisation: ``G`` groups of ``Ng`` points.  Test-mode inference at image scale s > 1/8 splits
the (5,h,w) lattice into r*r strided sub-lattices that the reference processes sequentially, each with
its own kNN and its own BatchNorm batch statistics (reference model.py:231-267); here they are the
groups of one batched launch (G = r*r, one stat group per group).  The nn.Module API of EdgeConv uses
G = batch size with all groups pooled into one stat group (BatchNorm2d pools over the batch).
"""
import contextlib
import ctypes

import torch

from . import _lib
from .utils.torch_utils import knn_lattice

_F32 = torch.float32


# ---------------------------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------------------------
def hip_inference(x, module):
    """True when ``module(x)`` needs no autograd graph and can take the HIP inference kernels: a float32 GPU tensor,
    and either grad mode off or nothing involved that requires grad.  (The reference's model.py calls the operator
    modules' plain ``forward``; this is how those calls reach the kernels.)"""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == _F32):
        return False
    if not torch.is_grad_enabled():
        return True
    return not (x.requires_grad or any(p.requires_grad for p in module.parameters()))


_LIBRARY_WARNED = set()


def warn_library_fallback(what, conv):
    """Say ONCE per layer shape that a convolution of the operator layer runs on the library (MIOpen through ATen)
    instead of this package's kernels: widths / kernel sizes the HIP kernels are not built for keep the reference's
    results but not the measured speed, and "no library kernel on the path" stops being true for that model."""
    key = (what, type(conv).__name__, conv.in_channels, conv.out_channels, tuple(conv.kernel_size), tuple(conv.stride))
    if key in _LIBRARY_WARNED:
        return
    _LIBRARY_WARNED.add(key)
    import warnings
    warnings.warn("pointmvsnet_amd: %s %s(%d -> %d, kernel %s, stride %s) is not a shape the HIP kernels are built for; "
                  "it runs on the library convolution (same results, not the measured speed)"
                  % (what, key[1], key[2], key[3], key[4], key[5]), RuntimeWarning, stacklevel=3)


def shared_mlp_supported(blocks):
    """The form shared_mlp_forward is built for: [Conv1d 1x1 (no bias) -> BatchNorm1d -> ReLU] blocks of GEMM widths."""
    for blk in blocks:
        conv, bn = blk.conv, blk.bn
        if (type(conv) is not torch.nn.Conv1d or conv.kernel_size != (1,) or conv.stride != (1,) or conv.groups != 1
                or conv.bias is not None or bn is None or not blk.relu
                or (conv.out_channels + 31) // 32 * 32 not in (32, 64, 128)       # pf_pointwise_gemm_f32's widths
                or bn.momentum is None or not bn.affine):
            return False
    return True


def shared_mlp_forward(blocks, x):
    """SharedMLP over points (reference nn/mlp.py:45-81 with ndim = 1: [Conv1d 1x1 -> BatchNorm1d -> ReLU] x n on
    (B, C, N)) on pf_pointwise_gemm_f32: the chain stays point-major, every BatchNorm + ReLU is applied by the next
    GEMM's A load, statistics pooled over the batch like BatchNorm1d.  Returns (B, C_out, N), or None when a block
    is not of that form (the caller then runs the stock composition)."""
    B, C, N = x.shape
    if not shared_mlp_supported(blocks):
        return None
    dev = x.device
    X, pm, ldx, K = x.contiguous(), False, 0, C
    affine = None
    with _lib.on_device(dev):
        for blk in blocks:
            Wt, cout = pack_weight_t(blk.conv.weight)
            Z = torch.empty((B * N, cout), dtype=_F32, device=dev)
            part = pointwise_gemm(X, pm, ldx, Wt, Z, cout, B, N, K, cout, in_affine=affine, groups_per_stat=B,
                                  want_stats=True)
            affine = _bn_affine_from_gemm(blk.bn, part, cout, B, N, B, dev, lazy=True)
            X, pm, ldx, K = Z, True, cout, cout
        scale, shift = affine_rows(affine)                 # (1, cout): the last BatchNorm + ReLU, then channel-major
        y = torch.relu_(X * scale + shift).view(B, N, K).transpose(1, 2).contiguous()
        flush_counters()
    return y


def stat_blocks(G, Ng):
    return int(_lib.load().pf_stat_blocks(int(G), int(Ng)))


# ---------------------------------------------------------------------------------------------
# packed-weight cache
# ---------------------------------------------------------------------------------------------
# An entry is valid only while every source tensor is the same OBJECT (weak reference), has the same version
# counter, the same storage address, device and dtype: ``param.data = ...``, ``module.to()/cuda(1)/double()``,
# ``vector_to_parameters`` and EMA swaps leave ``_version`` untouched but move the storage.  Entries a captured
# hipGraph reads are pinned (graph.GraphedForward): eviction skips them, so the allocator can never hand the
# memory a live graph reads to somebody else.
import collections as _collections
import weakref as _weakref

_PACK_CAP = 256
_pack_cache = _collections.OrderedDict()
_pack_pins = {}            # key -> pin count
_pack_log = None           # list collecting the keys used while a graph is being captured


def _src_sig(t):
    return (t._version, t.data_ptr(), str(t.device), t.dtype)


_pack_bypass = 0


class no_pack_cache(object):
    """Context: every packed-weight request re-packs (``make()``) instead of consulting the cache.  A hipGraph that
    contains a TRAINING step must hold the packing kernels themselves -- its parameters change between replays, so
    a pack cached at capture time would be stale at the first replay (train_step.GraphedTrainStep)."""

    def __enter__(self):
        global _pack_bypass
        _pack_bypass += 1
        return self

    def __exit__(self, *exc):
        global _pack_bypass
        _pack_bypass -= 1
        return False


# The packed weights of the training step in flight (train_packs.TrainPacks.active()): refreshed by one launch at the
# start of the step, found here under the same keys the wrappers below use.
_prepacked = None


def _cached_pack(key, sources, make):
    if _prepacked is not None:
        hit = _prepacked.get(key)
        if hit is not None:
            return hit
    if _pack_bypass:
        return make()
    hit = _pack_cache.get(key)
    if hit is not None:
        refs, sigs, out = hit
        if all(r() is t for r, t in zip(refs, sources)) and sigs == tuple(_src_sig(t) for t in sources):
            _pack_cache.move_to_end(key)
            if _pack_log is not None:
                _pack_log.append(key)
            return out
    out = make()
    try:
        _pack_cache[key] = (tuple(_weakref.ref(t) for t in sources), tuple(_src_sig(t) for t in sources), out)
        _pack_cache.move_to_end(key)
    except TypeError:
        return out
    if _pack_log is not None:
        _pack_log.append(key)
    if len(_pack_cache) > _PACK_CAP:                       # least recently used first, never a pinned entry
        for k in list(_pack_cache.keys()):
            if len(_pack_cache) <= _PACK_CAP:
                break
            if k not in _pack_pins and k != key:
                del _pack_cache[k]
    return out


def pack_log_begin():
    """Start recording which cache entries are used (graph capture)."""
    global _pack_log
    _pack_log = []


def pack_log_end(pin=True):
    """Stop recording; returns [(key, source weakrefs, signatures)] of the entries used, pinned against eviction."""
    global _pack_log
    keys, _pack_log = (_pack_log or []), None
    out = []
    for k in dict.fromkeys(keys):
        if k in _pack_cache:
            refs, sigs, _ = _pack_cache[k]
            out.append((k, refs, sigs))
            if pin:
                _pack_pins[k] = _pack_pins.get(k, 0) + 1
    return out


def pack_unpin(entries):
    for k, _, _ in entries:
        n = _pack_pins.get(k, 0) - 1
        if n <= 0:
            _pack_pins.pop(k, None)
        else:
            _pack_pins[k] = n


def pack_entries_stale(entries):
    """True when a source tensor of a recorded entry changed (value, storage, device or dtype) or died."""
    for _, refs, sigs in entries:
        for r, sig in zip(refs, sigs):
            t = r()
            if t is None or _src_sig(t) != sig:
                return True
    return False


def pack_weight_t(*convs):
    """Stack 1x1 conv weights (Cout_i, K, 1) along the output axis, transpose to (K, Nc) row-major and
    zero-pad Nc to a multiple of 32 (MFMA column tiles).  Returns (Wt, Cout_total).  Cached (see above)."""
    return _cached_pack(("wt",) + tuple(id(c) for c in convs), convs, lambda: _pack_weight_t(*convs))


def _pack_weight_t(*convs):
    w = torch.cat([c.reshape(c.shape[0], c.shape[1]) for c in convs], dim=0).detach().to(_F32)
    cout, K = w.shape
    nc = (cout + 31) // 32 * 32
    wt = torch.zeros((K, nc), dtype=_F32, device=w.device)
    wt[:, :cout] = w.t()
    return wt.contiguous(), cout


import os as _os

# PF_TIMELINE=1: one-thread timestamp kernels at the stage boundaries of PointMVSNet.run (they are graph nodes, so
# they work under hipGraph replay where events cannot); read back with timeline_report().
TIMELINE = int(_os.environ.get("PF_TIMELINE", "0"))
_timeline = {"buf": None, "names": []}


def stamp(name):
    if not TIMELINE:
        return
    t = _timeline
    if t["buf"] is None:
        t["buf"] = torch.zeros(64, dtype=torch.int64, device=torch.cuda.current_device())
    if name not in t["names"]:
        t["names"].append(name)
    i = t["names"].index(name)
    _lib.call("pf_debug_timestamp", _lib.ptr(t["buf"][i:i + 1]), _lib.stream())


def timeline_report():
    """{name: microseconds since the first stamp} of the LAST forward (call after a synchronize)."""
    t = _timeline
    if t["buf"] is None:
        return {}
    vals = t["buf"].cpu().tolist()
    t0 = min(vals[i] for i in range(len(t["names"])))
    return {n: (vals[i] - t0) / 100.0 for i, n in enumerate(t["names"])}


def stat_rows(G, T, pcols, dev):
    """Statistics rows of one launch: (G, T, pcols, 2) float64 per-block (sum, sum of squares)."""
    return torch.empty((G, T, pcols, 2), dtype=torch.float64, device=dev)


def pointwise_gemm(X, point_major, ldx, Wt, Y, ldy, G, Ng, K, nc_store, in_affine=None, groups_per_stat=1,
                   want_stats=False):
    """Y[:, :nc_store] = act(X) @ Wt (see pf_pointwise_gemm_f32).  Returns the float64 column partials
    (G, T, Nc, 2) when ``want_stats``.  ``in_affine``: (scale, shift) rows, or a LazyAffine (the pending BatchNorm
    is resolved by the GEMM's own blocks, no finalize launch)."""
    Nc = Wt.shape[1]
    T = int(_lib.load().pf_gemm_blocks(int(G), int(Ng)))
    partials = stat_rows(G, T, Nc, Wt.device) if want_stats else None
    sc, sh, in_bn = _split_affine(in_affine)
    _lib.call("pf_pointwise_gemm_f32",
              _lib.ptr(X), int(bool(point_major)), int(ldx), _lib.ptr(Wt), _lib.ptr(Y), int(ldy), int(G), int(Ng),
              int(K), int(Nc), int(nc_store), _lib.ptr(sc), _lib.ptr(sh), in_bn, int(groups_per_stat),
              _lib.ptr(partials), _lib.stream(),
              algo_bytes=4.0 * G * Ng * (K + nc_store) + 4.0 * K * Nc, flops=2.0 * G * Ng * K * nc_store)
    return partials


def bn_job(bn, partials, col0, C, count, unbias_n, G, groups_per_stat, scale, shift, ch0=0):
    """One ``pf_bn_job``: train-mode BatchNorm statistics -> (scale, shift) rows, plus the running-stat update
    the module would have made (one update per stat group, in order).  ``ch0`` selects the slice of the
    module's channels (EdgeConv's BN covers [central | difference])."""
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm momentum is not supported")
    T, pcols = partials.shape[1], partials.shape[2]
    track = bn.track_running_stats and bn.running_mean is not None
    dp = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    rm = bn.running_mean[ch0:ch0 + C] if track else None
    rv = bn.running_var[ch0:ch0 + C] if track else None
    return _lib.BnJob(dp(partials), int(T), int(pcols), int(col0), int(C), float(count), float(unbias_n),
                      dp(bn.weight.detach()[ch0:ch0 + C]), dp(bn.bias.detach()[ch0:ch0 + C]), dp(rm), dp(rv),
                      float(bn.momentum), float(bn.eps), int(G), int(groups_per_stat), dp(scale), dp(shift),
                      int(scale.stride(0)))


MAX_BN_JOBS = 32          # jobs per pf_bn_finalize_jobs_f32 launch (kBnJobs of csrc/edgeconv.hip)


def bn_finalize_jobs(jobs):
    """Up to MAX_BN_JOBS finalize jobs in one launch (pf_bn_finalize_jobs_f32)."""
    arr = (_lib.BnJob * len(jobs))(*jobs)
    _lib.call("pf_bn_finalize_jobs_f32", arr, len(jobs), _lib.stream(),
              algo_bytes=sum(16.0 * j.G * j.T * j.C for j in jobs))


def bn_affine(bn, partials, col0, C, count, unbias_n, G, groups_per_stat, scale, shift, ch0=0):
    """A single finalize job (see bn_job)."""
    bn_finalize_jobs([bn_job(bn, partials, col0, C, count, unbias_n, G, groups_per_stat, scale, shift, ch0)])


# PF_LAZY_BN=1 (default): a train-mode BatchNorm whose statistics rows are few (persistent GEMM blocks, the small
# maps of conv2d_wide) and whose consumer can resolve it (pf_bn_resolve, csrc/pf_bn_resolve.h) gets NO finalize launch
# on the critical path: the consumer's blocks compute (scale, shift) themselves, and the running statistics are
# updated by batched finalize launches on a side stream (flush_lazy_stats / join at flush_counters).
LAZY_BN = int(_os.environ.get("PF_LAZY_BN", "1"))
# What a consumer block is asked to re-reduce: rows behind one statistic x channels (16 bytes each).  Measured
# (profiles/archive/r02/r02aj_small_ab.txt): 4000 / 5120 / 16384 / 65536 -> 641 / 646 / 654 / 653 depth maps/s on config 2; the
# largest job there is the flow MLP at 25 600 points per group (200 rows x 64 channels = 205 KB per GEMM block).
LAZY_MAX_ELEMS = int(_os.environ.get("PF_LAZY_MAX", "16384"))


class LazyAffine(object):
    """A pending train-mode BatchNorm(+ReLU): the finished statistics rows of its producer and everything the
    finalize needs (one pf_bn_job).  Consumers with an ``in_bn`` slot take ``job_ptr()``; everybody else calls
    ``rows()``, which runs the ordinary finalize launch (once) and returns the (scale, shift) rows."""

    def __init__(self, job, keep, scale, shift):
        self.job, self.keep, self.scale, self.shift = job, keep, scale, shift
        self.done = False                 # finalize launched (rows valid, running statistics updated)
        self.queued = False

    def defer(self):
        """Queue the finalize for the next flush_lazy_stats (the rows and running statistics are wanted, just not now)."""
        if not (self.done or self.queued):
            self.queued = True
            _lazy_list(self.scale.device).append(self)

    def job_ptr(self):
        self.defer()
        return ctypes.pointer(self.job)

    def rows(self):
        if not self.done:
            bn_finalize_jobs([self.job])
            self.done = True
        return self.scale, self.shift


def _bn_tensors(bn):
    """Everything a pf_bn_job of ``bn`` points to: a queued LazyAffine keeps these alive until its deferred finalize
    has been launched (the job holds raw device pointers; the module may be gone by then)."""
    return tuple(t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None)


def _split_affine(in_affine):
    """(scale, shift, in_bn pointer) for a kernel with both kinds of input-affine slots."""
    if in_affine is None:
        return None, None, None
    if isinstance(in_affine, LazyAffine):
        if in_affine.done:
            return in_affine.scale, in_affine.shift, None
        return None, None, in_affine.job_ptr()
    return in_affine[0], in_affine[1], None


def affine_rows(in_affine):
    """(scale, shift) rows of ``in_affine`` (a tuple, a LazyAffine or None) for consumers without an in_bn slot."""
    if isinstance(in_affine, LazyAffine):
        return in_affine.rows()
    return in_affine


def _lazy_list(device):
    table = getattr(_pending, "lazy", None)
    if table is None:
        table = _pending.lazy = {}
    return table.setdefault(str(device), [])


def flush_lazy_stats(device=None):
    """Running statistics (and the scale/shift rows) of every BatchNorm that was resolved by its consumer since
    the last call: ONE batched finalize launch (<= MAX_BN_JOBS jobs each) on the current stream.  ``flush_counters`` calls
    it at the end of a forward -- the only place where it is off every consumer's critical path without a
    stream of its own (a further side stream made hipGraph serialise the coarse stage behind the flow tower:
    profiles/archive/r02/r02ah_lazy_bn_side_stream_timeline.txt)."""
    table = getattr(_pending, "lazy", {})
    for dev in list(table.keys()):
        if device is not None and str(device) != dev:
            continue
        lazy = [z for z in table[dev] if not z.done]
        del table[dev][:]
        # the jobs of one launch run concurrently: a module that was called several times (the flow MLP, once per
        # PointFlow iteration) gets its running-statistics updates in call order, one launch per call
        layers, seen = [], {}
        for z in lazy:
            k = seen.get(z.job.running_mean, 0) if z.job.running_mean else 0
            if z.job.running_mean:
                seen[z.job.running_mean] = k + 1
            while len(layers) <= k:
                layers.append([])
            layers[k].append(z)
        with _lib.on_device(torch.device(dev)):
            cur = None
            for z in lazy:                     # a job made on another stream (the training step's flow tower): its
                origin = getattr(z, "origin", None)        # tensors are read here, tell the caching allocator
                if origin is None:
                    continue
                cur = torch.cuda.current_stream() if cur is None else cur
                if origin != cur:
                    for t in z.keep:
                        if t.is_cuda:
                            t.record_stream(cur)
            for layer in layers:
                for i in range(0, len(layer), MAX_BN_JOBS):
                    bn_finalize_jobs([z.job for z in layer[i:i + MAX_BN_JOBS]])
        for z in lazy:
            z.done = True


# Concurrency level INSIDE one forward (PF_CONCURRENCY): 0 = single stream; 1 = flow tower beside the coarse stage;
# 2 = + lattice kNN beside the first EdgeConv GEMM (default: best for ONE scene at a time, 422 / 487 depth maps/s for
# 0 / 2 in round 1).  With several scenes in flight (graph.LanedForward) the lanes ARE the concurrency and every
# lane is captured as a single chain (level 0): forks inside the graphs cost hardware queues that the lanes need
# (profiles/archive/r03/r03b_lanes_queues.md: 3 lanes 771 / 938 / 808 depth maps/s at levels 0 / 1 / 2, 4 lanes on 4 queues 1030).
CONCURRENCY = int(_os.environ.get("PF_CONCURRENCY", "2"))


class concurrency(object):
    """Context: forwards issued (or captured) inside run at this intra-forward concurrency level."""

    def __init__(self, level):
        self.level = int(level)

    def __enter__(self):
        global CONCURRENCY
        self.saved, CONCURRENCY = CONCURRENCY, self.level
        return self

    def __exit__(self, *exc):
        global CONCURRENCY
        CONCURRENCY = self.saved
        return False

_side_streams = {}
_lane = 0


def set_lane(lane):
    """Scene lane of the forwards issued from now on (graph.GraphedForward with ``lanes`` > 1 captures several
    scenes side by side): every lane forks onto its OWN auxiliary streams, so one scene's flow tower never queues
    behind another scene's."""
    global _lane
    _lane = int(lane)


def current_lane():
    return _lane


def side_stream(device, slot):
    """A cached auxiliary stream (per device, scene lane and slot) for independent kernel chains; fork with
    ``s.wait_stream(current)``, join with ``current.wait_stream(s)`` -- graph edges under hipGraph capture."""
    key = (str(device), _lane, slot)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


# Deferred ``num_batches_tracked`` updates, per (thread, device): the reference trains under nn.DataParallel
# (train.py:177), which calls forward from one Python thread per GPU -- a shared list would let one replica
# flush another replica's counters (tensors of several devices in one _foreach_add_).
import threading as _threading

_pending = _threading.local()


def _pending_list(device):
    table = getattr(_pending, "table", None)
    if table is None:
        table = _pending.table = {}
    return table.setdefault(str(device), [])


def bump_counter(bn, n):
    """``num_batches_tracked += n`` deferred to one fused launch (flush_counters) instead of one tiny
    elementwise kernel per BatchNorm module (82 of them per depth map in eager PyTorch)."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        _pending_list(bn.num_batches_tracked.device).append((bn.num_batches_tracked, int(n)))


_defer_flush = [0]


@contextlib.contextmanager
def deferred_counters():
    """Inside: ``flush_counters()`` calls do nothing; ONE flush when the outermost context exits (the training forward:
    eight nodes each flushed their own BatchNorm counters, eight tiny launches in the step's chain)."""
    _defer_flush[0] += 1
    try:
        yield
    finally:
        _defer_flush[0] -= 1
        if _defer_flush[0] == 0:
            flush_counters()


def flush_counters():
    if _defer_flush[0] > 0:
        return
    flush_lazy_stats()
    for pending in list(getattr(_pending, "table", {}).values()):
        if pending:
            merged = {}
            for t, n in pending:                           # a module may be bumped several times per forward
                key = t.data_ptr()
                merged[key] = (t, merged[key][1] + n) if key in merged else (t, n)
            del pending[:]
            torch._foreach_add_([t for t, _ in merged.values()], [n for _, n in merged.values()])


def eval_affine(bn, S, ld, ch0=0, C=None, out=None):
    """Eval-mode BatchNorm (running statistics) folded to scale/shift rows."""
    C = bn.num_features - ch0 if C is None else C
    inv = torch.rsqrt(bn.running_var[ch0:ch0 + C] + bn.eps) * bn.weight.detach()[ch0:ch0 + C]
    sh = bn.bias.detach()[ch0:ch0 + C] - bn.running_mean[ch0:ch0 + C] * inv
    return inv, sh


# ---------------------------------------------------------------------------------------------
# BatchNorm (+ReLU) for the conv stacks around the path (ImageConv / VolumeConv)
# ---------------------------------------------------------------------------------------------
def pack_conv3d_weight_pair(weight):
    """(Cout<=8,Cin,3,3,3) -> (Cin/4, 36, 4, 16): column c + 8 s holds W[c] shifted by s along kh (see
    csrc/conv3d_pair.hip); cached per parameter."""
    def make():
        cout, cin = weight.shape[:2]
        w = weight.detach().to(_F32)
        wp = torch.zeros((cin // 4, 3, 4, 3, 4, 16), dtype=_F32, device=weight.device)   # g, kd, kh', kw, k, col
        src = w.permute(1, 2, 3, 4, 0).reshape(cin // 4, 4, 3, 3, 3, cout).permute(0, 2, 3, 4, 1, 5)   # g,kd,kh,kw,k,c
        for s_ in (0, 1):
            wp[:, :, s_:s_ + 3, :, :, 8 * s_:8 * s_ + cout] = src
        return wp.reshape(cin // 4, 36, 4, 16).contiguous()
    return _cached_pack(("c3p", id(weight)), (weight,), make)


def conv3d_k3(x, weight, stride, want_stats, in_affine=None, samples_per_stat=1):
    """3x3x3 / pad 1 conv3d on the f32 matrix cores (pf_conv3d_k3_f32; stride 1 with <= 8 output channels:
    pf_conv3d_k3_pair_f32).  ``in_affine``: the pending BatchNorm + ReLU of x -- (scale, shift) rows
    (N/samples_per_stat, Cin) or a LazyAffine -- applied while x is staged.  Returns (y, partials or None)."""
    N, Cin, Di, Hi, Wi = x.shape
    Cout = weight.shape[0]
    if stride == 1 and Cout <= 8 and Cin % 4 == 0 and in_affine is None:
        wp = pack_conv3d_weight_pair(weight)
        y = torch.empty((N, Cout, Di, Hi, Wi), dtype=_F32, device=x.device)
        partials = None
        if want_stats:
            T = int(_lib.load().pf_conv3d_pair_blocks(Cin, Cout, Di, Hi, Wi))
            partials = torch.empty((N, T, Cout, 2), dtype=torch.float64, device=x.device)
        _lib.call("pf_conv3d_k3_pair_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, Cout, Di, Hi, Wi,
                  _lib.ptr(partials), _lib.stream(),
                  algo_bytes=4.0 * N * (Cin + Cout) * Di * Hi * Wi + 4.0 * 27 * Cin * Cout,
                  flops=2.0 * N * Di * Hi * Wi * 27 * Cin * Cout)
        return y, partials
    Do, Ho, Wo = (Di - 1) // stride + 1, (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    wp = pack_conv3d_weight(weight)
    y = torch.empty((N, Cout, Do, Ho, Wo), dtype=_F32, device=x.device)
    partials = None
    if want_stats:
        T = int(_lib.load().pf_conv3d_blocks(Cin, Cout, Di, Hi, Wi, int(stride)))
        partials = torch.empty((N, T, Cout, 2), dtype=torch.float64, device=x.device)
    sc, sh, in_bn = _split_affine(in_affine)
    _lib.call("pf_conv3d_k3_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, Cout, Di, Hi, Wi, int(stride),
              _lib.ptr(sc), _lib.ptr(sh), in_bn, int(samples_per_stat), _lib.ptr(partials), _lib.stream(),
              algo_bytes=4.0 * N * (Cin * Di * Hi * Wi + Cout * Do * Ho * Wo) + 4.0 * 27 * Cin * Cout,
              flops=2.0 * N * Do * Ho * Wo * 27 * Cin * Cout)
    return y, partials


def deconv3d_k3s2(xa, xb, weight, want_stats, in_affine=None, samples_per_stat=1):
    """ConvTranspose3d 3x3x3 / stride 2 / padding 1 / output_padding 1 of ``act(xa) (+ xb)`` (pf_deconv3d_k3s2_f32;
    weight in nn.ConvTranspose3d's (Cin, Cout, 3, 3, 3) layout).  ``in_affine``: the pending BatchNorm + ReLU of
    xa (rows or a LazyAffine; None: xa as it is).  Returns (y, partials or None)."""
    N, Cin, D, H, W = xa.shape
    Cout = weight.shape[1]
    w = weight.detach()
    if w.dtype != _F32 or not w.is_contiguous():
        w = w.to(_F32).contiguous()
    y = torch.empty((N, Cout, 2 * D, 2 * H, 2 * W), dtype=_F32, device=xa.device)
    partials = None
    if want_stats:
        T = int(_lib.load().pf_deconv3d_blocks(D, H, W))
        partials = torch.empty((N, T, Cout, 2), dtype=torch.float64, device=xa.device)
    vol = D * H * W
    sc, sh, in_bn = _split_affine(in_affine)
    _lib.call("pf_deconv3d_k3s2_f32", _lib.ptr(xa), _lib.ptr(xb), _lib.ptr(w), _lib.ptr(y), N, Cin, Cout, D, H, W,
              _lib.ptr(sc), _lib.ptr(sh), in_bn, int(samples_per_stat), _lib.ptr(partials), _lib.stream(),
              algo_bytes=4.0 * N * vol * (Cin * (2 if xb is not None else 1) + 8 * Cout) + 4.0 * 27 * Cin * Cout,
              flops=2.0 * N * vol * 27 * Cin * Cout)
    return y, partials


def conv2d_supported(conv):
    """True for the two conv shapes of the feature towers that pf_conv2d_f32 implements."""
    if type(conv) is not torch.nn.Conv2d or conv.bias is not None or conv.groups != 1 or conv.out_channels > 64:
        return False
    if conv.dilation != (1, 1):
        return False
    k3 = conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
    k5 = conv.kernel_size == (5, 5) and conv.stride == (2, 2) and conv.padding == (2, 2)
    return k3 or k5


def conv2d_wide_supported(conv):
    """Shapes pf_conv2d_wide_f32 is built for: 3x3/1 3->8, 3->16 (two towers' first layers stacked), 8->8, 16->16,
    32->32, 64->64; 5x5/2 8->16, 16->32, 32->64."""
    return conv2d_supported(conv) and bool(_lib.load().pf_conv2d_wide_supported(
        conv.in_channels, conv.out_channels, int(conv.kernel_size[0]), int(conv.stride[0])))


def conv2d_wide_preferred(conv):
    """Measured (profiles/archive/r02/r02ag_microbench_conv2d_wide.log, cfg2 shapes, 3 views; us): 16->32 5x5/2 25.1 (pf_conv2d_f32
    31.0, library 34.7), 32->32 3x3 16.7 (22.8, 24.4), 32->64 5x5/2 21.0 (40.5, 31.9), 64->64 3x3 16.5 (38.0, 24.6) --
    62-76 TF of exact f32: every 32- and 64-channel tower layer runs on csrc/conv2d_wide.hip."""
    return conv2d_wide_supported(conv)


def _pack_conv2d_wide(weight):
    cout, cin, k, _ = weight.shape
    if cout == 8 and cin >= 8 and _os.environ.get("PF_WIDE_PAIR", "1") != "0":      # (0: tools' A/B against a round-3 library)
        # paired rows (csrc/conv2d_wide.hip, Wide16Cfg::PAIR): column co + 8 s of the MFMA tile is channel co of output
        # row 2 rp + s, which takes row tap kh' = kh + s of the K + 1 patch rows the pair reads
        cinp = (cin + 3) // 4 * 4
        full = torch.zeros((k + 1, k, cinp, 2, 8), dtype=_F32, device=weight.device)      # [kh'][kw][ci][s][co]
        src = weight.detach().to(_F32).permute(2, 3, 1, 0)                                    # [kh][kw][ci][co]
        for s_ in (0, 1):
            full[s_:s_ + k, :, :cin, s_, :cout] = src
        return full.view(k + 1, k, 4, cinp // 4, 16).permute(0, 1, 2, 4, 3).contiguous()
    if cout <= 16:
        cinp = (cin + 3) // 4 * 4
        full = torch.zeros((k, k, cinp, 16), dtype=_F32, device=weight.device)
        full[:, :, :cin, :cout] = weight.detach().to(_F32).permute(2, 3, 1, 0)
        return full.view(k, k, 4, cinp // 4, 16).permute(0, 1, 2, 4, 3).contiguous()
    w = weight.detach().to(_F32).permute(2, 3, 1, 0).reshape(k, k, cin // 8, 2, 4, cout)
    return w.permute(0, 1, 2, 3, 5, 4).contiguous()


def pack_conv2d_wide_weight(weight):
    """(Cout,Cin,K,K) -> (K, K, Cin/8, 2, Cout, 4): [kh][kw][kc][h][co][j] = w[co][8 kc + 4 h + j][kh][kw]
    (Cout 32 / 64: 32x32x2 MFMA), or, for Cout = 8 / 16 (16x16x4 MFMA), (K, K, 4, 16, Cin'/4) with Cin' = Cin
    rounded up to 4: [kh][kw][kq][co][j] = w[co][(Cin'/4) kq + j][kh][kw], zero where co >= Cout or the channel
    does not exist.  Cached per weight (see the packed-weight cache above)."""
    return _cached_pack(("c2w", id(weight)), (weight,), lambda: _pack_conv2d_wide(weight))


def conv2d_wide(x, conv, in_affine, samples_per_stat, want_stats, channel_last_out=False):
    """pf_conv2d_wide_f32: same contract as ``conv2d`` (raw y, statistics partials or None); ``in_affine`` may be
    a LazyAffine (resolved by the launch's own blocks).  ``channel_last_out``: y is (N, Ho, Wo, Cout)."""
    N, Cin, Hi, Wi = x.shape
    Cout = conv.out_channels
    ks, stride = conv.kernel_size[0], conv.stride[0]
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    wp = pack_conv2d_wide_weight(conv.weight)
    y = torch.empty((N, Ho, Wo, Cout) if channel_last_out else (N, Cout, Ho, Wo), dtype=_F32, device=x.device)
    partials = None
    if want_stats:
        T = int(_lib.load().pf_conv2d_wide_blocks(Cout, Hi, Wi, int(stride)))
        partials = stat_rows(N, T, Cout, x.device)
    sc, sh, in_bn = _split_affine(in_affine)
    _lib.call("pf_conv2d_wide_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, Cout, Hi, Wi, int(ks), int(stride),
              _lib.ptr(sc), _lib.ptr(sh), in_bn, int(samples_per_stat), _lib.ptr(partials), int(bool(channel_last_out)),
              _lib.stream(), algo_bytes=4.0 * N * (Cin * Hi * Wi + Cout * Ho * Wo) + 4.0 * ks * ks * Cin * Cout,
              flops=2.0 * N * Ho * Wo * ks * ks * Cin * Cout,
              tag="%d->%d %dx%d/%d" % (Cin, Cout, ks, ks, stride))
    return y, partials


class AffineSets(object):
    """The pending BatchNorm(+ReLU)s of a launch over parameter sets (the model's two towers side by side): joint
    (sets * G, C) scale / shift rows -- set s owns rows [s G, (s + 1) G) -- and, while nobody has asked for the rows,
    one LazyAffine per set whose consumer resolves it (pf_conv2d_wide_sets_f32's ``in_bn`` array)."""

    def __init__(self, scale, shift, sets, lazies=None):
        self.scale, self.shift, self.sets, self.lazies = scale, shift, int(sets), lazies

    def split(self):
        """(scale, shift, in_bn array) for the consuming launch."""
        if self.lazies is not None and not any(l.done for l in self.lazies):
            for l in self.lazies:
                l.job_ptr()                     # (queues the deferred finalize: running statistics, rows)
            return None, None, (_lib.BnJob * len(self.lazies))(*[l.job for l in self.lazies])
        self.rows()
        return self.scale, self.shift, None

    def rows(self):
        if self.lazies is not None:
            todo = [l for l in self.lazies if not l.done]
            if todo:
                bn_finalize_jobs([l.job for l in todo])
                for l in todo:
                    l.done = True
        return self.scale, self.shift

    def rows_of(self, s):
        sc, sh = self.rows()
        G = sc.shape[0] // self.sets
        return sc[s * G:(s + 1) * G], sh[s * G:(s + 1) * G]


def pack_conv2d_wide_weight_sets(weights):
    """The packed weights of ``pack_conv2d_wide_weight`` for several same-shaped convolutions, stacked (sets, ...)."""
    def make():
        return torch.stack([_pack_conv2d_wide(w) for w in weights]).contiguous()
    return _cached_pack(("c2ws",) + tuple(id(w) for w in weights), tuple(weights), make)


def conv2d_wide_stacked_supported(convs):
    c = convs[0]
    return all(conv2d_supported(k) for k in convs) and bool(_lib.load().pf_conv2d_wide_supported(
        c.in_channels, sum(k.out_channels for k in convs), int(c.kernel_size[0]), int(c.stride[0])))


def conv2d_wide_stacked(x, convs, want_stats):
    """The towers' FIRST layer: same input, so the sets' output channels are stacked into one convolution (3 -> 8 + 8
    fills all 16 columns of the matrix tile).  Returns y (n, sets * Cout, Ho, Wo) -- sample-major, set-interleaved:
    the ``interleaved`` input layout of conv2d_wide_sets -- and the statistics partials (n, T, sets * Cout, 2)."""
    N, Cin, Hi, Wi = x.shape
    Cout = sum(c.out_channels for c in convs)
    ks, stride = convs[0].kernel_size[0], convs[0].stride[0]
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    weights = tuple(c.weight for c in convs)
    wp = _cached_pack(("c2wst",) + tuple(id(w) for w in weights), weights,
                      lambda: _pack_conv2d_wide(torch.cat([w.detach() for w in weights], dim=0)))
    y = torch.empty((N, Cout, Ho, Wo), dtype=_F32, device=x.device)
    partials = None
    if want_stats:
        T = int(_lib.load().pf_conv2d_wide_blocks(Cout, Hi, Wi, int(stride)))
        partials = stat_rows(N, T, Cout, x.device)
    _lib.call("pf_conv2d_wide_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, Cout, Hi, Wi, int(ks), int(stride),
              None, None, None, 1, _lib.ptr(partials), 0, _lib.stream(),
              algo_bytes=4.0 * N * (Cin * Hi * Wi + Cout * Ho * Wo) + 4.0 * ks * ks * Cin * Cout,
              flops=2.0 * N * Ho * Wo * ks * ks * Cin * Cout,
              tag="%d->%d %dx%d/%d" % (Cin, Cout, ks, ks, stride))
    return y, partials


def conv2d_wide_sets(x, convs, in_affine, samples_per_stat, want_stats, interleaved=False, channel_last_sets=()):
    """pf_conv2d_wide_sets_f32: ONE launch for the same layer of several towers.  x: (sets * n, Cin, H, W) -- set s owns
    samples [s n, (s + 1) n) -- or, ``interleaved``, (n, sets * Cin, H, W) as conv2d_wide_stacked wrote it;
    ``in_affine``: None or an AffineSets; ``channel_last_sets``: the sets whose samples are written (Ho, Wo, Cout).
    Returns (y (sets * n, Cout, Ho, Wo), partials)."""
    sets = len(convs)
    conv = convs[0]
    if interleaved:
        x = x.view(x.shape[0] * sets, x.shape[1] // sets, x.shape[2], x.shape[3])
    Cin, Hi, Wi = x.shape[1:]
    N = x.shape[0]
    Cout = conv.out_channels
    ks, stride = conv.kernel_size[0], conv.stride[0]
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    wp = pack_conv2d_wide_weight_sets([c.weight for c in convs])
    y = torch.empty((N, Cout, Ho, Wo), dtype=_F32, device=x.device)
    partials = None
    if want_stats:
        T = int(_lib.load().pf_conv2d_wide_blocks(Cout, Hi, Wi, int(stride)))
        partials = stat_rows(N, T, Cout, x.device)
    sc, sh, in_bn = (None, None, None) if in_affine is None else in_affine.split()
    mask = sum(1 << s for s in channel_last_sets)
    _lib.call("pf_conv2d_wide_sets_f32", _lib.ptr(x), 2 if interleaved else 0, _lib.ptr(wp), int(wp[0].numel()), sets,
              _lib.ptr(y), N, Cin, Cout, Hi, Wi, int(ks), int(stride), _lib.ptr(sc), _lib.ptr(sh), in_bn,
              int(samples_per_stat), _lib.ptr(partials), int(mask), _lib.stream(),
              algo_bytes=4.0 * (x.numel() + N * Cout * Ho * Wo) + 4.0 * sets * ks * ks * Cin * Cout,
              flops=2.0 * N * Ho * Wo * ks * ks * Cin * Cout,
              tag="%d->%d %dx%d/%d" % (Cin, Cout, ks, ks, stride))
    return y, partials


def bn_affine_rows_sets(y, bns, samples_per_stat, partials, lazy, interleaved=False):
    """``bn_affine_rows`` for the raw output y (sets * n, C, h, w) of a conv2d_wide_sets launch in train mode: one job per
    set over its own slice of the statistics rows, joint (sets * G, C) affine rows.  ``interleaved``: y is
    (n, sets * C, h, w) from conv2d_wide_stacked -- a set's job reads its C columns of every row.  Returns an AffineSets."""
    sets = len(bns)
    S = y[0, 0].numel()
    n, C = (y.shape[0], y.shape[1] // sets) if interleaved else (y.shape[0] // sets, y.shape[1])
    G = n // samples_per_stat
    scale = torch.empty((sets * G, C), dtype=_F32, device=y.device)
    shift = torch.empty((sets * G, C), dtype=_F32, device=y.device)
    cnt = float(samples_per_stat) * S
    jobs = []
    for s, bn in enumerate(bns):
        rows, col0 = (partials, s * C) if interleaved else (partials[s * n:(s + 1) * n], 0)
        jobs.append(bn_job(bn, rows, col0, C, cnt, cnt, n, samples_per_stat,
                           scale[s * G:(s + 1) * G], shift[s * G:(s + 1) * G]))
        bump_counter(bn, G)
    if lazy and LAZY_BN and samples_per_stat * partials.shape[1] * C <= LAZY_MAX_ELEMS:
        lazies = [LazyAffine(j, (partials, scale, shift) + _bn_tensors(bn), scale, shift) for j, bn in zip(jobs, bns)]
        return AffineSets(scale, shift, sets, lazies)
    bn_finalize_jobs(jobs)
    return AffineSets(scale, shift, sets)


def channel_affine_(x, affine, relu, samples_per_stat):
    """In place y = act(x*scale + shift) with (N/sps, C) affine rows."""
    affine = affine_rows(affine)
    N, C = x.shape[:2]
    S = x[0, 0].numel()
    _lib.call("pf_channel_affine_f32", _lib.ptr(x), _lib.ptr(x), _lib.ptr(affine[0]), _lib.ptr(affine[1]), N, C, S,
              int(samples_per_stat), int(bool(relu)), _lib.stream(), algo_bytes=8.0 * N * C * S)
    return x


# one-launch BatchNorm (pf_channel_bn_fused_f32) when a channel's N*S elements fit one block's streaming budget
FUSED_BN_MAX = 65536


def _bn_fused(x, y, bn, relu, samples_per_stat, scale=None, shift=None):
    N, C = x.shape[:2]
    S = x[0, 0].numel()
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm momentum is not supported")
    track = bn.track_running_stats and bn.running_mean is not None
    _lib.call("pf_channel_bn_fused_f32", _lib.ptr(x), _lib.ptr(y), N, C, S, int(samples_per_stat),
              _lib.ptr(bn.weight.detach()), _lib.ptr(bn.bias.detach()), _lib.ptr(bn.running_mean if track else None),
              _lib.ptr(bn.running_var if track else None), float(bn.momentum), float(bn.eps), int(bool(relu)),
              _lib.ptr(scale), _lib.ptr(shift), int(scale.stride(0)) if scale is not None else 0, _lib.stream(),
              algo_bytes=(8.0 if y is not None else 4.0) * N * C * S)


def bn_affine_rows(x, bn, samples_per_stat, partials=None, lazy=False):
    """(scale, shift) rows (N/sps, C) of BatchNorm ``bn`` for the raw conv output x (N,C,*spatial): batch
    statistics from ``partials`` (or a statistics pass over x) in train mode, running statistics in eval.
    ``lazy``: the consumer has an ``in_bn`` slot -- with few statistics rows a LazyAffine is returned instead."""
    N, C = x.shape[:2]
    S = x[0, 0].numel()
    G = N // samples_per_stat
    dev = x.device
    if bn.training or not bn.track_running_stats:
        scale = torch.empty((G, C), dtype=_F32, device=dev)
        shift = torch.empty((G, C), dtype=_F32, device=dev)
        if partials is None and N * S <= FUSED_BN_MAX:
            _bn_fused(x, None, bn, False, samples_per_stat, scale, shift)
            bump_counter(bn, G)
            return scale, shift
        if partials is None:
            T = int(_lib.load().pf_norm_blocks(S))
            partials = torch.empty((N, T, C, 2), dtype=torch.float64, device=dev)
            _lib.call("pf_channel_stats_f32", _lib.ptr(x), N, C, S, _lib.ptr(partials), _lib.stream(),
                      algo_bytes=4.0 * N * C * S)
        n = float(samples_per_stat) * S
        bump_counter(bn, G)
        if lazy and LAZY_BN and samples_per_stat * partials.shape[1] * C <= LAZY_MAX_ELEMS:
            job = bn_job(bn, partials, 0, C, n, n, N, samples_per_stat, scale, shift)
            return LazyAffine(job, (partials, scale, shift) + _bn_tensors(bn), scale, shift)
        bn_affine(bn, partials, 0, C, n, n, N, samples_per_stat, scale, shift)
        return scale, shift
    sc, sh = eval_affine(bn, G, C)
    return sc.unsqueeze(0).expand(G, C).contiguous(), sh.unsqueeze(0).expand(G, C).contiguous()


def conv3d_bottom_supported(conv):
    """nn.Conv3d shapes of pf_conv3d_bottom_f32: 3x3x3 / pad 1, 32 -> 64 stride 2 or 64 -> 64 stride 1, no bias."""
    return (type(conv) is torch.nn.Conv3d and conv.kernel_size == (3, 3, 3) and conv.padding == (1, 1, 1)
            and conv.stride in ((1, 1, 1), (2, 2, 2)) and conv.dilation == (1, 1, 1) and conv.groups == 1
            and conv.bias is None and bool(_lib.load().pf_conv3d_bottom_supported(
                conv.in_channels, conv.out_channels, int(conv.stride[0]))))


def deconv3d_bottom_supported(conv):
    """nn.ConvTranspose3d shape of pf_deconv3d_bottom_f32: 3x3x3, stride 2, pad 1, output_padding 1, 64 -> 32."""
    return (type(conv) is torch.nn.ConvTranspose3d and conv.kernel_size == (3, 3, 3) and conv.stride == (2, 2, 2)
            and conv.padding == (1, 1, 1) and conv.output_padding == (1, 1, 1) and conv.dilation == (1, 1, 1)
            and conv.groups == 1 and conv.bias is None
            and bool(_lib.load().pf_deconv3d_bottom_supported(conv.in_channels, conv.out_channels)))


def pack_conv3d_bottom_weight(weight):
    """(64,Cin,3,3,3) -> (3,3,3,Cin/16,4,64,4): [kd][kh][kw][kc][kq][co][j] = w[co][16 kc + 4 kq + j][kd][kh][kw]."""
    def make():
        cout, cin = weight.shape[:2]
        w = weight.detach().to(_F32).permute(2, 3, 4, 1, 0).reshape(3, 3, 3, cin // 16, 4, 4, cout)
        return w.permute(0, 1, 2, 3, 4, 6, 5).contiguous()
    return _cached_pack(("c3b", id(weight)), (weight,), make)


def pack_deconv3d_bottom_weight(weight):
    """ConvTranspose3d weight (64,32,3,3,3) -> (27,4,4,32,4): [tap][kc][kq][co][j] = w[16 kc + 4 kq + j][co][tap]."""
    def make():
        cin, cout = weight.shape[:2]
        w = weight.detach().to(_F32).permute(2, 3, 4, 0, 1).reshape(27, cin // 16, 4, 4, cout)
        return w.permute(0, 1, 2, 4, 3).contiguous()
    return _cached_pack(("d3b", id(weight)), (weight,), make)


def conv3d_bottom(x, conv, in_affine, samples_per_stat, want_stats):
    """pf_conv3d_bottom_f32 -> (raw y, statistics partials (N, T, 64, 2) or None); ``in_affine``: (scale, shift)
    rows, a LazyAffine (resolved by the launch) or None."""
    N, Cin, Di, Hi, Wi = x.shape
    stride = int(conv.stride[0])
    Do, Ho, Wo = (Di - 1) // stride + 1, (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    wp = pack_conv3d_bottom_weight(conv.weight)
    y = torch.empty((N, 64, Do, Ho, Wo), dtype=_F32, device=x.device)
    partials = None
    if want_stats:
        T = int(_lib.load().pf_conv3d_bottom_blocks(Di, Hi, Wi, stride))
        partials = torch.empty((N, T, 64, 2), dtype=torch.float64, device=x.device)
    sc, sh, in_bn = _split_affine(in_affine)
    _lib.call("pf_conv3d_bottom_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, 64, Di, Hi, Wi, stride,
              _lib.ptr(sc), _lib.ptr(sh), in_bn, int(samples_per_stat), _lib.ptr(partials), _lib.stream(),
              algo_bytes=4.0 * N * (Cin * Di * Hi * Wi + 64 * Do * Ho * Wo) + 4.0 * 27 * Cin * 64,
              flops=2.0 * N * Do * Ho * Wo * 27 * Cin * 64)
    return y, partials


def deconv3d_bottom(x, conv, in_affine, samples_per_stat, want_stats):
    """pf_deconv3d_bottom_f32 -> (raw y (N,32,2D,2H,2W), statistics partials (N, T, 32, 2) or None)."""
    N, Cin, Di, Hi, Wi = x.shape
    wp = pack_deconv3d_bottom_weight(conv.weight)
    y = torch.empty((N, 32, 2 * Di, 2 * Hi, 2 * Wi), dtype=_F32, device=x.device)
    partials = None
    if want_stats:
        T = int(_lib.load().pf_deconv3d_bottom_blocks(Di, Hi, Wi))
        partials = torch.empty((N, T, 32, 2), dtype=torch.float64, device=x.device)
    sc, sh, in_bn = _split_affine(in_affine)
    _lib.call("pf_deconv3d_bottom_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, 32, Di, Hi, Wi,
              _lib.ptr(sc), _lib.ptr(sh), in_bn, int(samples_per_stat), _lib.ptr(partials), _lib.stream(),
              algo_bytes=4.0 * N * Di * Hi * Wi * (Cin + 8 * 32) + 4.0 * 27 * Cin * 32,
              flops=2.0 * N * Di * Hi * Wi * 27 * Cin * 32)
    return y, partials


def conv3d_k3_few(x, weight):
    """3x3x3 / pad 1 / stride 1 conv3d with <= 4 output channels (pf_conv3d_k3_few_f32)."""
    N, Cin, D, H, W = x.shape
    Cout = weight.shape[0]
    y = torch.empty((N, Cout, D, H, W), dtype=_F32, device=x.device)
    w = weight.detach().to(_F32).contiguous()
    _lib.call("pf_conv3d_k3_few_f32", _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), N, Cin, Cout, D, H, W, _lib.stream(),
              algo_bytes=4.0 * N * D * H * W * (Cin + Cout))
    return y


def pack_conv3d_weight(weight):
    """(Cout,Cin,3,3,3) -> (Cin/4, 27, 4, 16*ceil(Cout/16)) zero padded; cached per parameter."""
    def make():
        cout, cin = weight.shape[:2]
        ncp = (cout + 15) // 16 * 16
        wp = torch.zeros((cin // 4, 27, 4, ncp), dtype=_F32, device=weight.device)
        wp[..., :cout] = weight.detach().to(_F32).permute(1, 2, 3, 4, 0).reshape(cin // 4, 4, 27, cout).transpose(1, 2)
        return wp
    return _cached_pack(("c3", id(weight)), (weight,), make)


def batch_norm_act_(x, bn, relu, samples_per_stat, partials=None, addend=None):
    """In-place train/eval BatchNorm + optional ReLU on a contiguous (N,C,*spatial) conv output; with
    ``addend`` (same shape) the result is ``addend + act(bn(x))`` (a decoder skip add in the same pass).

    ``samples_per_stat`` consecutive samples share one set of batch statistics == one reference module
    call (the reference runs each view through the tower separately, model.py:71-77, so views batched
    along N keep per-view statistics and the running statistics are updated once per view, in order).
    ``partials``: per-block statistics already produced by the convolution's epilogue (N, T, C, 2)."""
    N, C = x.shape[:2]
    S = x[0, 0].numel()
    G = N // samples_per_stat
    dev = x.device
    if bn.training or not bn.track_running_stats:
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm momentum is not supported")
        if partials is None and N * S <= FUSED_BN_MAX:
            _bn_fused(x, x, bn, relu, samples_per_stat)
            bump_counter(bn, G)
            return x if addend is None else x.add_(addend)
        if partials is None:
            T = int(_lib.load().pf_norm_blocks(S))
            partials = torch.empty((N, T, C, 2), dtype=torch.float64, device=dev)
            _lib.call("pf_channel_stats_f32", _lib.ptr(x), N, C, S, _lib.ptr(partials), _lib.stream(),
                      algo_bytes=4.0 * N * C * S)
        track = bn.track_running_stats and bn.running_mean is not None
        if addend is not None and (addend.shape != x.shape or not addend.is_contiguous()):
            raise RuntimeError("batch_norm_act_: addend must be contiguous and shaped like x")
        _lib.call("pf_channel_bn_apply_f32", _lib.ptr(x), _lib.ptr(x), _lib.ptr(partials), int(partials.shape[1]),
                  N, C, S, int(samples_per_stat), float(samples_per_stat) * S, _lib.ptr(bn.weight.detach()),
                  _lib.ptr(bn.bias.detach()), _lib.ptr(bn.running_mean if track else None),
                  _lib.ptr(bn.running_var if track else None), float(bn.momentum), float(bn.eps), int(bool(relu)),
                  _lib.ptr(addend), _lib.stream(), algo_bytes=(8.0 if addend is None else 12.0) * N * C * S)
        bump_counter(bn, G)
        return x
    else:
        sc, sh = eval_affine(bn, G, C)
        scale = sc.unsqueeze(0).expand(G, C).contiguous()
        shift = sh.unsqueeze(0).expand(G, C).contiguous()
        _lib.call("pf_channel_affine_f32", _lib.ptr(x), _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), N, C, S,
                  int(samples_per_stat), int(bool(relu)), _lib.stream(), algo_bytes=8.0 * N * C * S)
    return x if addend is None else x.add_(addend)


def batch_norm_act2_(x1, bn1, partials1, x2, bn2, partials2, samples_per_stat):
    """relu(bn2(x2)) + relu(bn1(x1)) in ONE pass, written into x1 (pf_channel_bn_apply2_f32): both inputs are raw
    convolution outputs with their statistics partials, both BatchNorms in train mode with ReLU (VolumeConv's last
    skip add, reference networks.py:166)."""
    N, C = x1.shape[:2]
    S = x1[0, 0].numel()
    if x2.shape != x1.shape or not (x1.is_contiguous() and x2.is_contiguous()):
        raise RuntimeError("batch_norm_act2_: the two tensors must be contiguous and of one shape")
    for bn in (bn1, bn2):
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm momentum is not supported")
    def side(bn):
        track = bn.track_running_stats and bn.running_mean is not None
        return (_lib.ptr(bn.weight.detach()), _lib.ptr(bn.bias.detach()), _lib.ptr(bn.running_mean if track else None),
                _lib.ptr(bn.running_var if track else None), float(bn.momentum), float(bn.eps))
    a, b = side(bn1), side(bn2)
    _lib.call("pf_channel_bn_apply2_f32", _lib.ptr(x1), _lib.ptr(partials1), int(partials1.shape[1]), *a,
              _lib.ptr(x2), _lib.ptr(partials2), int(partials2.shape[1]), *b, _lib.ptr(x1), N, C, S,
              int(samples_per_stat), float(samples_per_stat) * S, _lib.stream(), algo_bytes=12.0 * N * C * S)
    G = N // samples_per_stat
    bump_counter(bn1, G)
    bump_counter(bn2, G)
    return x1


# ---------------------------------------------------------------------------------------------
# EdgeConv (rows E0 / E1 / E2)
# ---------------------------------------------------------------------------------------------
def edge_conv_fused(X, point_major, ldx, K, G, Ng, idx, conv1_w, conv2_w, bn, concat, Y, ldy,
                    groups_per_stat=1, join=None, codes=None, lattice=None, keep=None):
    """One EdgeConv / EdgeConvNoC layer on G groups of Ng points (reference networks.py:18-45, :56-81).

    X: channel-major (G,K,Ng) or point-major rows; idx (G,Ng,k) int64 group-local; Y: point-major view
    with ``ldy`` floats per point receiving [central | diff] (concat) or diff (NoC).  Alternatively to ``idx``:
    ``codes`` (G,Ng,16) uint8 window codes of the lattice kNN with ``lattice`` = (window, H, W).
    ``keep`` (a dict) receives what the backward pass recomputes from: the rows LE = [l | e], the BatchNorm
    affine rows and the batch statistics (mean, invstd) in the BatchNorm's channel order."""
    C = conv1_w.shape[0]
    k = idx.shape[-1] if idx is not None else int(codes.shape[-1])
    # (without codes a lattice (0, H, W) is only a hint for the XCD-aware tile order of the two gather passes)
    lat = tuple(int(v) for v in lattice) if lattice is not None else (0, 1, 1)
    if codes is None:
        lat = (0, lat[1], lat[2])
    if keep is not None:
        keep["plane"] = lat[1] * lat[2] if lat[1] * lat[2] > 1 else 0
    dev = Y.device
    S = G // groups_per_stat
    training = bn.training or not bn.track_running_stats
    Wt, _ = pack_weight_t(conv1_w, conv2_w)                      # (K, 2C): [l | e]
    LE = torch.empty((G * Ng, 2 * C), dtype=_F32, device=dev)
    cbn = 2 * C if concat else C
    scale = torch.empty((S, cbn), dtype=_F32, device=dev)
    shift = torch.empty((S, cbn), dtype=_F32, device=dev)
    n_pairs = float(groups_per_stat) * Ng * k
    part_l = pointwise_gemm(X, point_major, ldx, Wt, LE, 2 * C, G, Ng, K, 2 * C, groups_per_stat=groups_per_stat,
                            want_stats=(concat and training))
    if join is not None:                      # ``idx`` was produced on another stream (flow_chain)
        torch.cuda.current_stream().wait_stream(join)
    rows4 = None
    if training:
        T = stat_blocks(G, Ng)
        part_d = stat_rows(G, T, C, dev)
        _lib.call("pf_edge_stats_f32", _lib.ptr(LE), 2 * C, C, _lib.ptr(idx), k, G, Ng, _lib.ptr(part_d),
                  _lib.ptr(codes), lat[0], lat[1], lat[2], _lib.stream(),
                  algo_bytes=float(G) * Ng * (4.0 * C + (1.0 if codes is not None else 8.0) * k + 4.0 * C * k))
        if keep is not None:
            # training: one finalize per half that also keeps (mean, invstd) for the backward (pf_bn_train_rows_f32)
            from . import train_ops
            rows4 = torch.empty((4, S, cbn), dtype=_F32, device=dev)

            def job4(partials, count, unbias_n, col, ch0):
                j = bn_job(bn, partials, 0, C, count, unbias_n, G, groups_per_stat, rows4[0][:, col:], rows4[1][:, col:],
                           ch0=ch0)
                j.rows4 = 1                # (rows 2 / 3 = mean / invstd, at the same columns)
                return j

            if not train_ops.TRAIN_LAZY_BN:          # round 4's form: one pf_bn_train_rows_f32 launch per half
                if concat:
                    train_ops.bn_train_rows(bn, part_l, 0, C, float(groups_per_stat) * Ng, G, groups_per_stat,
                                            unbias_n=n_pairs, rows=rows4, col_out=0, ch0=0, bump=False)
                    train_ops.bn_train_rows(bn, part_d, 0, C, n_pairs, G, groups_per_stat, rows=rows4, col_out=C, ch0=C,
                                            bump=False)
                else:
                    train_ops.bn_train_rows(bn, part_d, 0, C, n_pairs, G, groups_per_stat, rows=rows4, bump=False)
            elif concat:                   # central and difference halves: separate statistics, ONE launch
                bn_finalize_jobs([job4(part_l, float(groups_per_stat) * Ng, n_pairs, 0, 0),
                                  job4(part_d, n_pairs, n_pairs, C, C)])
            else:
                bn_finalize_jobs([job4(part_d, n_pairs, n_pairs, 0, 0)])
            scale, shift = rows4[0], rows4[1]
        elif concat:
            bn_finalize_jobs([
                bn_job(bn, part_l, 0, C, float(groups_per_stat) * Ng, n_pairs, G, groups_per_stat, scale, shift),
                bn_job(bn, part_d, 0, C, n_pairs, n_pairs, G, groups_per_stat, scale[:, C:], shift[:, C:], ch0=C)])
        else:
            bn_affine(bn, part_d, 0, C, n_pairs, n_pairs, G, groups_per_stat, scale, shift, ch0=0)
        bump_counter(bn, S)
    else:
        sc, sh = eval_affine(bn, S, cbn)
        scale.copy_(sc.unsqueeze(0).expand(S, cbn))
        shift.copy_(sh.unsqueeze(0).expand(S, cbn))
    if keep is not None:
        if not training:
            raise RuntimeError("edge_conv_fused(keep=...) needs a train-mode BatchNorm")
        keep.update(LE=LE, scale=rows4[0], shift=rows4[1], mean=rows4[2], invstd=rows4[3])
    _lib.call("pf_edge_apply_f32", _lib.ptr(LE), 2 * C, C, _lib.ptr(idx), k, G, Ng, _lib.ptr(scale),
              _lib.ptr(shift), cbn, groups_per_stat, int(bool(concat)), _lib.ptr(Y), int(ldy), _lib.ptr(codes),
              lat[0], lat[1], lat[2], _lib.stream(),
              algo_bytes=float(G) * Ng * (4.0 * C + (1.0 if codes is not None else 8.0) * k + 4.0 * C * k + 4.0 * cbn))
    return Y


# The inverted index tensors of the current backward pass: the three EdgeConv layers of a PointFlow iteration share
# one idx, so the sort runs once per iteration (keyed by the tensor's storage and version; a handful of entries).
_inverse_cache = _collections.OrderedDict()


def knn_inverse(idx, G, Ng, k):
    """(order, start) of pf_knn_inverse for idx (G, Ng, k) int64 (contiguous): who gathers each point, in pair order."""
    key = (idx.data_ptr(), idx._version, str(idx.device), G, Ng, k)
    hit = _inverse_cache.get(key)
    if hit is not None and hit[0]() is idx:
        return hit[1], hit[2]
    dev = idx.device
    order = torch.empty((G * Ng * k,), dtype=torch.int32, device=dev)
    start = torch.empty((G * Ng + 1,), dtype=torch.int32, device=dev)
    nbytes = int(_lib.load().pf_knn_inverse_workspace(int(G), int(Ng), int(k)))
    work = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
    _lib.call("pf_knn_inverse", _lib.ptr(idx), int(k), int(G), int(Ng), _lib.ptr(order), _lib.ptr(start), _lib.ptr(work),
              nbytes, _lib.stream(), algo_bytes=8.0 * G * Ng * k * 4)
    try:
        _inverse_cache[key] = (_weakref.ref(idx), order, start)
    except TypeError:
        return order, start
    while len(_inverse_cache) > 8:
        _inverse_cache.popitem(last=False)
    return order, start


# True: the de rows of the EdgeConv backward are gathered over the inverted index lists (bit-reproducible); False: the
# reference's float atomics (module attribute for the tests that compare the two, not an environment switch)
DETERMINISTIC_BACKWARD = True
# PF_EDGE_BWD_SUMS=0: round 4's three walks (reduce, apply over the forward lists, gather over the inverted lists)
# instead of two (reduce leaves the per-point sums, the inverted-list gather finishes dl as well)
EDGE_BWD_SUMS = _os.environ.get("PF_EDGE_BWD_SUMS", "1") != "0"


def edge_conv_backward(keep, idx, grad_y, C, k, G, Ng, groups_per_stat, concat, into=None, grad_acc=None):
    """Gradient of edge_conv_fused's output rows w.r.t. LE = [l | e] and the BatchNorm affine parameters
    (pf_edge_backward_reduce_f32 / _coeffs_f32 / _apply_f32: d = e[idx] - l is recomputed, nothing of size N*k is
    stored).  grad_y: (G*Ng, cbn) point-major.  Returns (grad_LE (G*Ng, 2C), grad_gamma (cbn,), grad_beta (cbn,));
    ``into`` = (dgamma, dbeta) tensors to ADD the parameter gradients to (then the returned ones are None).
    ``grad_acc``: a second upstream gradient (G*Ng, cbn), contiguous and OWNED by the caller -- the gradient is
    grad_y + grad_acc; the two-walk form adds them inside its first pass (grad_acc then holds the sum)."""
    LE, scale, shift, mean, invstd = keep["LE"], keep["scale"], keep["shift"], keep["mean"], keep["invstd"]
    dev = LE.device
    cbn = 2 * C if concat else C
    S = G // groups_per_stat
    T = stat_blocks(G, Ng)
    partials = torch.empty((G, T, cbn, 2), dtype=torch.float64, device=dev)
    grad_le = torch.empty((G * Ng, 2 * C), dtype=_F32, device=dev)
    two_walks = DETERMINISTIC_BACKWARD and EDGE_BWD_SUMS
    if grad_acc is not None and not two_walks:
        grad_y, grad_acc = grad_y + grad_acc, None
    if two_walks:
        # the reduce pass leaves every point's own (sum g, sum xhat) in grad_le: dl is linear in the coefficients, so
        # the finish pass (the gather over the inverted lists) completes it without walking the forward lists again
        _lib.call("pf_edge_backward_sums_f32", _lib.ptr(LE), 2 * C, C, _lib.ptr(idx), k, G, Ng, _lib.ptr(grad_y),
                  int(grad_y.stride(0)), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), cbn,
                  groups_per_stat, int(bool(concat)), _lib.ptr(partials), _lib.ptr(grad_le), _lib.ptr(grad_acc),
                  0 if grad_acc is None else int(grad_acc.stride(0)), int(keep.get("plane", 0)), _lib.stream(),
                  algo_bytes=float(G) * Ng * (12.0 * C + 8.0 * k + 4.0 * C * k + 4.0 * cbn))
        if grad_acc is not None:
            grad_y = grad_acc
    else:
        _lib.call("pf_edge_backward_reduce_f32", _lib.ptr(LE), 2 * C, C, _lib.ptr(idx), k, G, Ng, _lib.ptr(grad_y),
                  int(grad_y.stride(0)), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), cbn,
                  groups_per_stat, int(bool(concat)), _lib.ptr(partials), _lib.stream(),
                  algo_bytes=float(G) * Ng * (4.0 * C + 8.0 * k + 4.0 * C * k + 4.0 * cbn))
    c1 = torch.empty((S, cbn), dtype=_F32, device=dev)
    c2 = torch.empty((S, cbn), dtype=_F32, device=dev)
    if into is None:
        dgamma = torch.empty((cbn,), dtype=_F32, device=dev)
        dbeta = torch.empty((cbn,), dtype=_F32, device=dev)
    else:
        dgamma, dbeta = into
    _lib.call("pf_edge_backward_coeffs_f32", _lib.ptr(partials), G, T, cbn, C, int(bool(concat)), groups_per_stat, Ng, k,
              _lib.ptr(c1), _lib.ptr(c2), _lib.ptr(dgamma), _lib.ptr(dbeta), 0 if into is None else 1, _lib.stream(),
              algo_bytes=16.0 * G * T * cbn)
    order, start = knn_inverse(idx, G, Ng, k) if DETERMINISTIC_BACKWARD else (None, None)
    if two_walks:
        _lib.call("pf_edge_backward_finish_f32", _lib.ptr(LE), 2 * C, C, k, G, Ng, _lib.ptr(grad_y),
                  int(grad_y.stride(0)), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(c1),
                  _lib.ptr(c2), cbn, groups_per_stat, int(bool(concat)), _lib.ptr(grad_le), _lib.ptr(order),
                  _lib.ptr(start), int(keep.get("plane", 0)), _lib.stream(),
                  algo_bytes=float(G) * Ng * (16.0 * C + 4.0 * k + 8.0 * C * k + 4.0 * cbn))
    else:
        _lib.call("pf_edge_backward_apply_f32", _lib.ptr(LE), 2 * C, C, _lib.ptr(idx), k, G, Ng, _lib.ptr(grad_y),
                  int(grad_y.stride(0)), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(c1),
                  _lib.ptr(c2), cbn, groups_per_stat, int(bool(concat)), _lib.ptr(grad_le), _lib.ptr(order), _lib.ptr(start),
                  _lib.stream(), algo_bytes=float(G) * Ng * (8.0 * C + 8.0 * k + 8.0 * C * k + 4.0 * cbn))
    return (grad_le, dgamma, dbeta) if into is None else (grad_le, None, None)


def _bn_affine_from_gemm(bn, partials, C, G, Ng, groups_per_stat, dev, lazy=False):
    """BatchNorm1d after a pointwise GEMM: statistics over the points of a stat group.  ``lazy``: the caller's
    next kernel has an ``in_bn`` slot -- return a LazyAffine instead of launching the finalize (train mode, few
    statistics rows)."""
    S = G // groups_per_stat
    scale = torch.empty((S, C), dtype=_F32, device=dev)
    shift = torch.empty((S, C), dtype=_F32, device=dev)
    if bn.training or not bn.track_running_stats:
        n = float(groups_per_stat) * Ng
        bump_counter(bn, S)
        if lazy and LAZY_BN and groups_per_stat * partials.shape[1] * C <= LAZY_MAX_ELEMS:
            job = bn_job(bn, partials, 0, C, n, n, G, groups_per_stat, scale, shift)
            return LazyAffine(job, (partials, scale, shift) + _bn_tensors(bn), scale, shift)
        bn_affine(bn, partials, 0, C, n, n, G, groups_per_stat, scale, shift)
    else:
        sc, sh = eval_affine(bn, S, C)
        scale.copy_(sc.unsqueeze(0).expand(S, C))
        shift.copy_(sh.unsqueeze(0).expand(S, C))
    return scale, shift


# ---------------------------------------------------------------------------------------------
# rows F, K, E*, M, H, T for one scene: one PointFlow iteration
# ---------------------------------------------------------------------------------------------
def resize_maps(maps, h, w):
    """(V,C,IH,IW) -> (V,C,h,w), bilinear align_corners=False (the F.interpolate of model.py:184)."""
    V, C, IH, IW = maps.shape
    if IH == h and IW == w:
        return maps
    out = torch.empty((V, C, h, w), dtype=_F32, device=maps.device)
    _lib.call("pf_resize_bilinear_f32", _lib.ptr(maps), _lib.ptr(out), V * C, IH, IW, h, w, _lib.stream(),
              algo_bytes=4.0 * V * C * (IH * IW + h * w))
    return out


class RawLevel(object):
    """A pyramid level as the tower's last convolution left it: ``raw`` (V, c, h_l, w_l) with its BatchNorm + ReLU still
    pending -- ``affine`` = (scale, shift) rows (V, c), or a LazyAffine.  flow_pyramid applies it while it resizes."""

    def __init__(self, raw, affine):
        self.raw, self.affine = raw, affine


def flow_pyramid(pyramid, h, w):
    """The three pyramid levels of a scene resized to the flow grid (model.py:180-186) in one launch, CHANNEL-LAST
    (V,h,w,c_l) -- the layout flow_features samples with 16-byte loads.  A level is a (V,c_l,h_l,w_l) tensor, or a
    ``RawLevel`` whose pending BatchNorm + ReLU the kernel applies to every texel before interpolating."""
    maps = [(m.raw if isinstance(m, RawLevel) else m) for m in pyramid]
    rows = [(affine_rows(m.affine) if isinstance(m, RawLevel) else None) for m in pyramid]
    V = maps[0].shape[0]
    outs = [torch.empty((V, h, w, int(m.shape[1])), dtype=_F32, device=m.device) for m in maps]
    args = []
    for m in maps:
        args += [_lib.ptr(m), int(m.shape[1]), int(m.shape[2]), int(m.shape[3])]
    sc = sh = None
    if any(r is not None for r in rows):
        sc = (ctypes.c_void_p * 3)(*[None if r is None else r[0].data_ptr() for r in rows])
        sh = (ctypes.c_void_p * 3)(*[None if r is None else r[1].data_ptr() for r in rows])
    _lib.call("pf_flow_pyramid_f32", *args, V, h, w, _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]), sc, sh,
              _lib.stream(), algo_bytes=sum(4.0 * m.numel() + 4.0 * o.numel() for m, o in zip(maps, outs)))
    return outs


def flow_features(levels, depth, interval, cam, h, w, ratio):
    """Row F.  levels: three channel-last (V,h,w,c) maps (flow_pyramid); depth (dh,dw); interval: 1-element
    device tensor; cam: packed camera block (device float32).  Returns feature (G, Ng, 136) -- point-major
    rows -- and xyz (G, 3, Ng), points in sub-grid-major order."""
    V = levels[0].shape[0]
    c1, c2, c3 = (int(l.shape[3]) for l in levels)
    G = ratio * ratio
    Ng = 5 * (h // ratio) * (w // ratio)
    dev = depth.device
    feature = torch.empty((G, Ng, c1 + c2 + c3 + 24), dtype=_F32, device=dev)
    xyz = torch.empty((G, 3, Ng), dtype=_F32, device=dev)
    _lib.call("pf_flow_features_f32",
              _lib.ptr(levels[0]), _lib.ptr(levels[1]), _lib.ptr(levels[2]), c1, c2, c3, V, h, w, _lib.ptr(depth),
              int(depth.shape[-2]), int(depth.shape[-1]), _lib.ptr(interval), _lib.ptr(cam), int(ratio),
              _lib.ptr(feature), _lib.ptr(xyz), _lib.stream(),
              algo_bytes=4.0 * V * (c1 + c2 + c3) * h * w + 4.0 * G * Ng * (c1 + c2 + c3 + 24 + 3) + 4.0 * h * w)
    return feature, xyz


def flow_chain(feature, xyz, depth, interval, h, w, ratio, edge_convs, flow_mlp, k=16, point_major=True, idx=None):
    """Rows K, E0-E2, M, H, T on assembled point features: feature (G,Ng,136) point-major rows (or
    (G,136,Ng) with ``point_major=False``) / xyz (G,3,Ng), points in sub-grid-major order (see
    flow_features) -> (depth_out (h,w), flow_prob (5,h,w)).  ``idx`` (G,Ng,k) int64 group-local neighbour
    indices replaces the lattice kNN of ``xyz`` (stage tests feed the oracle's own indices so that the
    EdgeConv / MLP / head chain is compared on identical neighbour sets)."""
    dev = depth.device
    if point_major:
        G, Ng, Cin = feature.shape
    else:
        G, Cin, Ng = feature.shape
    hs, ws = h // ratio, w // ratio
    # the lattice kNN needs only xyz; the first EdgeConv GEMM needs only the features: run them concurrently
    aux, codes, lattice = None, None, None
    if idx is not None:
        if tuple(idx.shape) != (G, Ng, k) or idx.dtype != torch.int64:
            raise RuntimeError("flow_chain: idx must be int64 (G, Ng, k)")
        idx = idx.contiguous()
    else:
        # window codes (16 bytes per point) instead of int64 indices when the kernels support it
        use_codes = k == 16                  # 16-byte window codes instead of 128 bytes of int64 indices per point
        lattice = (5, hs, ws) if use_codes else None

        def _knn():
            out = knn_lattice(xyz.view(G, 3, 5, hs, ws), 5, k, with_codes=use_codes, with_idx=not use_codes)
            return out if use_codes else (out, None)

        if CONCURRENCY >= 2:
            main = torch.cuda.current_stream()
            aux = side_stream(dev, 1)
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                idx, codes = _knn()                                        # (G, Ng, k), group-local
                (codes if use_codes else idx).record_stream(main)
        else:
            idx, codes = _knn()

    widths = []
    for m in edge_convs:
        c = m.conv1.weight.shape[0]
        widths.append(2 * c if m.concat else c)
    ctot = sum(widths)
    edges = torch.empty((G * Ng, ctot), dtype=_F32, device=dev)            # the (N,224) concat buffer
    col = 0
    X, pm, ldx, K = feature, bool(point_major), (Cin if point_major else 0), Cin
    for li, (m, wdt) in enumerate(zip(edge_convs, widths)):
        Y = edges[:, col:]
        edge_conv_fused(X, pm, ldx, K, G, Ng, idx, m.conv1.weight, m.conv2.weight, m.bn, m.concat, Y, ctot,
                        join=(aux if li == 0 else None), codes=codes, lattice=lattice)
        X, pm, ldx, K = Y, True, ctot, wdt
        col += wdt

    # flow MLP: SharedMLP (conv1d + BN1d + ReLU) x3, then Conv1d 16->1 (model.py:40-43)
    shared, last = flow_mlp[0], flow_mlp[1]
    X, ldx, K = edges, ctot, ctot
    affine = None
    for layer in shared:
        Wt, cout = pack_weight_t(layer.conv.weight)
        Z = torch.empty((G * Ng, cout), dtype=_F32, device=dev)
        bn = layer.bn
        part = pointwise_gemm(X, True, ldx, Wt, Z, cout, G, Ng, K, cout, in_affine=affine, want_stats=True)
        affine = _bn_affine_from_gemm(bn, part, cout, G, Ng, 1, dev, lazy=True)
        X, ldx, K = Z, cout, cout
    if K != 16 or last.weight.shape[0] != 1:
        raise NotImplementedError("flow head kernel is built for the reference widths (..., 16, 1)")
    depth_out = torch.empty((h, w), dtype=_F32, device=dev)
    flow_prob = torch.empty((5, h, w), dtype=_F32, device=dev)
    w_out = last.weight.detach().reshape(-1).to(_F32).contiguous()
    sc, sh, in_bn = _split_affine(affine)
    _lib.call("pf_flow_head_f32", _lib.ptr(X), ldx, _lib.ptr(sc), _lib.ptr(sh), 16, in_bn,
              _lib.ptr(w_out), _lib.ptr(depth), int(depth.shape[-2]), int(depth.shape[-1]), _lib.ptr(interval), h, w,
              ratio, _lib.ptr(flow_prob), _lib.ptr(depth_out), _lib.stream(),
              algo_bytes=4.0 * G * Ng * 16 + 4.0 * h * w * 7)
    return depth_out, flow_prob


def flow_iteration(pyramid, depth, interval, cam, h, w, ratio, edge_convs, flow_mlp, k=16):
    """One PointFlow refinement of one scene (reference model.py:150-295 for batch item b).

    pyramid: three contiguous (V,c,H_l,W_l) feature maps of this scene; depth: (dh,dw) prior depth map;
    interval: 1-element device tensor (hypothesis spacing); cam: packed camera block for this scale.
    Returns (depth_out (h,w), flow_prob (5,h,w))."""
    levels = flow_pyramid(pyramid, h, w)
    feature, xyz = flow_features(levels, depth, interval, cam, h, w, ratio)
    return flow_chain(feature, xyz, depth, interval, h, w, ratio, edge_convs, flow_mlp, k=k)


def soft_argmin_prob(cost, depth_start, depth_end, depth_interval):
    """Row S: cost (B,D,H,W) filtered volume -> depth (B,1,H,W), prob (B,1,H,W).
    depth_start/end/interval: (B,) float32 tensors on the same device."""
    return soft_argmin_params(cost, torch.stack([depth_start, depth_end, depth_interval], dim=1).to(_F32))


def soft_argmin_params(cost, params):
    """Row S with params (B,3) = (depth_start, depth_end, depth_interval) already on the device."""
    B, D, H, W = cost.shape
    cost = cost.contiguous()
    params = params.contiguous()
    depth = torch.empty((B, 1, H, W), dtype=_F32, device=cost.device)
    prob = torch.empty((B, 1, H, W), dtype=_F32, device=cost.device)
    _lib.call("pf_softargmin_prob_f32", _lib.ptr(cost), _lib.ptr(params), _lib.ptr(depth), _lib.ptr(prob),
              B, D, H * W, _lib.stream(), algo_bytes=4.0 * B * H * W * (D + 2))
    return depth, prob
