"""Row Z -- the training step of BASELINE config 4 on this package's own kernels (reference train.py:72-82).

The reference differentiates its conv -> BatchNorm -> ReLU stacks (networks.py:84-167 through nn/conv.py:62-77,
:108-121, :197-210; the flow MLP, model.py:40-43, through nn/conv.py:24-35) with ATen's convolution_backward and
batch_norm backward.  Here every such stack is ONE ``torch.autograd.Function`` whose forward runs the fused inference
kernels (raw convolution output + BatchNorm statistics from the epilogue, the pending BatchNorm + ReLU applied by the
next convolution while it stages its input) and whose backward is hand-written on the C ABI:

    BatchNorm+ReLU backward   pf_bn_bwd_reduce + pf_bn_bwd_apply_fused (csrc/norm_bwd.hip; rows forms for the MLP)
    weight gradients          pf_conv_wgrad_f32 / pf_rows_wgrad_f32 (csrc/conv_wgrad.hip; f32 MFMA, fixed-order sums;
                              a node's layers reduced by ONE pf_wgrad_reduce_batch_f32)
    data gradients            stride-1 layers: the FORWARD kernel on the flipped, transposed weight;
                              stride-2 convolutions: the transposed-convolution kernels (pf_deconv2d_k5s2_f32,
                              pf_deconv3d_k3s2_f32, pf_deconv3d_bottom_f32); ConvTranspose3d layers: the stride-2
                              forward convolution on the weight read as (Cout', Cin') = (Cin, Cout)
    1x1 convolutions          pf_pointwise_gemm_f32 on W itself (dX = dY W)
    warps, resizes            csrc/warp_bwd.hip: the bilinear scatters as gathers over sorted lists
    soft argmin, flow head,   csrc/train_heads.hip: one launch per direction each (autograd's compositions were 15-25
    masked MAE loss           element-wise launches)

No library convolution, BatchNorm or GEMM kernel runs in the step, and nothing uses float atomics: the gradient is
bit-reproducible.  One scene per process (B = 1, the reference's per-replica batch under DataParallel with 8 scenes on
8 GPUs, train.py:177): other batch sizes, other widths and eval-mode BatchNorm take the composed ATen path.
"""
import torch

from . import _lib
from . import pointflow

import contextlib
import ctypes
import os

_F32 = torch.float32

# The step's packed weights (train_packs.TrainPacks) while a fused training forward / backward runs; None: every
# function below packs on the fly with torch operators (the operator tests, eager use of single nodes).
_PACKS = None


@contextlib.contextmanager
def use_packs(packs):
    global _PACKS
    saved, _PACKS = _PACKS, packs
    try:
        if packs is None:
            yield None
        else:
            with packs.active():
                yield packs
    finally:
        _PACKS = saved


def _packed(kind, tensor):
    return None if _PACKS is None else _PACKS.get(kind, tensor)


# True (train_step.TrainStep sets it around forward + backward): a node adds its parameter gradients straight into the
# parameters' ``.grad`` tensors -- the views of the flat all-reduce bucket (distributed.GradBucket), zeroed at the start
# of the step -- from inside its own kernels (their accumulate flags) and hands autograd None for them, instead of
# returning 115 tensors that autograd then adds with one element-wise launch each (160 launches, 0.56 ms per step).
DIRECT_GRADS = False


@contextlib.contextmanager
def direct_grads(on=True):
    global DIRECT_GRADS
    saved, DIRECT_GRADS = DIRECT_GRADS, bool(on)
    if on:                                  # (a step that died in its backward must not leave launches for the next one)
        _MAIN["stream"] = torch.cuda.current_stream() if torch.cuda.is_available() else None
        del _WGRAD_DEFERRED[:]
        del _LATE["wgrad"][:]
        del _LATE["reduce"][:]
    try:
        yield
    finally:
        DIRECT_GRADS = saved


def _grad_target(p):
    """``p.grad`` when a node may accumulate into it directly, else None."""
    if not DIRECT_GRADS:
        return None
    g = p.grad
    if g is None or g.dtype != _F32 or g.shape != p.shape or not g.is_contiguous() or g.requires_grad:
        return None
    return g


# The training forward's second stream (model.TRAIN_FORK): set by the model while a fused training forward that forked
# the flow tower is being built, read by the nodes whose backward has work that only the flow tower's backward consumes
# (level 2: the pyramid-level gradients of _FlowFeaturesTrain).  None: everything on the node's own stream.
_SIDE = {"stream": None, "level": 0}


@contextlib.contextmanager
def side_stream(stream, level):
    saved = dict(_SIDE)
    _SIDE["stream"], _SIDE["level"] = stream, int(level)
    try:
        yield
    finally:
        _SIDE.update(saved)


def _on_side(min_level):
    return _SIDE["stream"] if _SIDE["level"] >= min_level else None


# (Weight gradients have no consumer inside the step either, but streams of their own for them -- per layer, per node,
# or the flow tower's stream -- were measured to LOSE 2-5 % beside the flow-tower fork: profiles/r04c_train_streams.md;
# the variants are in git history, not here.)

# Weight gradients that are ADDED into the bucket (direct_grads) leave their split partials in the workspace and queue the
# reduction; a node's backward reduces all of its layers' partials in one launch when it returns
# (pf_wgrad_reduce_batch_f32: the same fixed order per layer, 45 launches of ~5 us in the chain become 7).
# PF_WGRAD_BATCH=0: one reduce launch per layer.
WGRAD_BATCH = int(os.environ.get("PF_WGRAD_BATCH", "1"))
_REDUCE_PENDING = []


def _queue_reduce(work, into, elems, nbytes, rows, taps=1, swapped=False):
    """``rows`` / ``taps``: the Cg and the tap count of the launch that wrote ``work`` (its partials are (splits, rows, taps,
    columns)); ``swapped``: a swapped-operand launch (_conv_wgrad_swapped)."""
    late = (WGRAD_DEFER and WGRAD_LATE >= 2 and _MAIN["stream"] is not None
            and torch.cuda.current_stream(work.device) == _MAIN["stream"])
    (_LATE["reduce"] if late else _REDUCE_PENDING).append((work, into, int(elems), int(nbytes // (4 * elems)), int(rows), int(taps), int(bool(swapped)),
                            torch.cuda.current_stream(work.device)))


# Convolution weight gradients that are ADDED into the bucket are not even launched where they are computed: a node queues
# them and issues them together when its backward returns (pf_conv_wgrad_batch_f32: layers that share a kernel instantiation
# ride in ONE grid -- VolumeConv's 96-384-block layers of ~20 us each beside conv1_0's), then the one batched reduction.
# PF_WGRAD_DEFER=0: every layer launched in place (round 5's order).
WGRAD_DEFER = int(os.environ.get("PF_WGRAD_DEFER", "1"))
# ... and the 1x1 layers of the PointFlow nodes (EdgeConv chain, MLP: point-major rows) wait even longer: until the END of the
# backward (flush_late(), called by model.join_fork_streams()), where the 25 600-point iteration's six launches of 10-30 us
# ride in the grids of the 102 400-point iteration's (6.21 -> 6.10 ms per step).  2 (default): the convolution layers of the
# nodes that run on the step's main stream (coarse tower, VolumeConv) wait for the end too (6.10 -> 6.07 ms); the flow tower's,
# on its own stream, are issued there when its node returns.  PF_WGRAD_LATE=0: everything flushed with its own node.
WGRAD_LATE = int(os.environ.get("PF_WGRAD_LATE", "2"))
_WGRAD_DEFERRED = []
_LATE = {"wgrad": [], "reduce": []}


_MAIN = {"stream": None}


def _defer_wgrad(gr, x, N, Cg, Cx, go, xi, k3, stride, p3, sc, sh, sps, work, nbytes, flops, algo_bytes):
    late = WGRAD_LATE >= 2 and _MAIN["stream"] is not None and torch.cuda.current_stream(gr.device) == _MAIN["stream"]
    (_LATE["wgrad"] if late else _WGRAD_DEFERRED).append(dict(gr=gr, x=x, N=int(N), Cg=int(Cg), Cx=int(Cx), go=tuple(int(v) for v in go),
                                xi=tuple(int(v) for v in xi), k3=tuple(int(v) for v in k3), stride=int(stride),
                                p3=tuple(int(v) for v in p3), sc=sc, sh=sh, sps=int(sps), work=work, nbytes=int(nbytes),
                                flops=float(flops), bytes=float(algo_bytes), rows=None))


def _wgrad_launch_items(todo):
    items = (_lib.WgradItem * len(todo))()
    for it, d in zip(items, todo):
        it.gr, it.x = d["gr"].data_ptr(), d["x"].data_ptr()
        it.Cg, it.Cx = d["Cg"], d["Cx"]
        it.x_scale = None if d["sc"] is None else d["sc"].data_ptr()
        it.x_shift = None if d["sh"] is None else d["sh"].data_ptr()
        it.workspace, it.workspace_bytes = d["work"].data_ptr(), d["nbytes"]
        if d["rows"] is not None:
            it.rows_P, it.ldg, it.ldx, it.x_rows_per_stat = d["rows"]
            it.stride = 1
            continue
        it.N = d["N"]
        it.Do, it.Ho, it.Wo = d["go"]
        it.Di, it.Hi, it.Wi = d["xi"]
        it.KD, it.KH, it.KW = d["k3"]
        it.stride = d["stride"]
        it.pd, it.ph, it.pw = d["p3"]
        it.x_samples_per_stat = d["sps"]
    with torch.cuda.device(todo[0]["gr"].device):
        for base in range(0, len(todo), 64):                      # (the entry point takes <= 64 items)
            n = min(64, len(todo) - base)
            chunk = (_lib.WgradItem * n).from_address(ctypes.addressof(items) + base * ctypes.sizeof(_lib.WgradItem))
            _lib.call("pf_conv_wgrad_batch_f32", chunk, n, _lib.stream(),
                      algo_bytes=sum(d["bytes"] for d in todo[base:base + n]),
                      flops=sum(d["flops"] for d in todo[base:base + n]))


def _wgrad_flush_deferred():
    if not _WGRAD_DEFERRED:
        return
    todo = list(_WGRAD_DEFERRED)
    del _WGRAD_DEFERRED[:]
    _wgrad_launch_items(todo)


# The late weight gradients go out on PF_WGRAD_STREAMS streams (default 2: the caller's and a fork stream; 1 = one stream),
# dealt by their flops: each is a chip-filling grid of 85-240 us that drains for a good part of its run time, and nothing
# orders one layer's weight gradient against another's -- on two streams the drain of one overlaps the next one's start
# (cfg-4 step 5.60 -> 5.50 ms, same box, three alternations).
WGRAD_STREAMS = int(os.environ.get("PF_WGRAD_STREAMS", "2"))
_EXTRA_STREAMS = {}


def _wgrad_streams(side, n):
    """``n`` - 1 streams beside the current one: the fork stream first, then streams of this module's own."""
    out = [side]
    key = str(side.device)
    pool = _EXTRA_STREAMS.setdefault(key, [])
    while len(pool) < n - 2:
        pool.append(torch.cuda.Stream(device=side.device))
    return out + pool[:max(0, n - 2)]


def flush_late(side=None):
    """The end of a backward (model.join_fork_streams()): the weight gradients that waited for it, then their reduction."""
    todo, reds = list(_LATE["wgrad"]), list(_LATE["reduce"])
    del _LATE["wgrad"][:]
    del _LATE["reduce"][:]
    if todo and WGRAD_STREAMS >= 2 and side is not None and len(todo) >= 2:
        streams = _wgrad_streams(side, WGRAD_STREAMS)
        groups = [[] for _ in range(len(streams) + 1)]       # groups[0]: the current stream
        load = [0.0] * len(groups)
        for i in sorted(range(len(todo)), key=lambda i: -todo[i].get("flops", 0.0)):
            g = load.index(min(load))                        # greedy: the next largest to the lightest stream
            groups[g].append(todo[i])
            load[g] += todo[i].get("flops", 0.0)
        cur = torch.cuda.current_stream(side.device)
        for st, grp in zip(streams, groups[1:]):
            if not grp:
                continue
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                _wgrad_launch_items(grp)
                for d in grp:
                    d["work"].record_stream(st)
        if groups[0]:
            _wgrad_launch_items(groups[0])
        for st, grp in zip(streams, groups[1:]):
            if grp:
                cur.wait_stream(st)
        # (the reductions stay on ONE stream, in queue order: two PointFlow iterations add into the same parameters'
        # gradients -- on two streams that is a race, and it was measured to buy nothing: 5.50 -> 5.49 ms)
    elif todo:
        _wgrad_launch_items(todo)
    if reds:
        with torch.cuda.device(reds[0][0].device):
            for base in range(0, len(reds), 16):
                _reduce_launch(reds[base:base + 16])


def _reduce_flush():
    """One pf_wgrad_reduce_batch_f32 launch per stream the node's weight gradients were issued on (normally one: the
    current stream; with model.TRAIN_FORK = 3 the PointFlow nodes' weight gradients and their reduction run on the side
    stream, where nothing of the chain waits for them)."""
    _wgrad_flush_deferred()
    if not _REDUCE_PENDING:
        return
    everything = list(_REDUCE_PENDING)
    del _REDUCE_PENDING[:]
    streams = []
    for p in everything:
        if not any(p[7] == st for st in streams):
            streams.append(p[7])
    for st in streams:
        with torch.cuda.stream(st):
            _reduce_launch([p for p in everything if p[7] == st])


def _reduce_launch(pending):
    n = len(pending)
    parts = (ctypes.c_void_p * n)(*[p[0].data_ptr() for p in pending])
    dws = (ctypes.c_void_p * n)(*[p[1].data_ptr() for p in pending])
    elems = (ctypes.c_int64 * n)(*[p[2] for p in pending])
    splits = (ctypes.c_int * n)(*[p[3] for p in pending])
    rows = (ctypes.c_int * n)(*[p[4] for p in pending])
    taps = (ctypes.c_int * n)(*[p[5] for p in pending])
    swapped = (ctypes.c_int * n)(*[p[6] for p in pending])
    with torch.cuda.device(pending[0][0].device):
        cur = torch.cuda.current_stream()
        for p in pending:                      # (a partial may have been written on the side stream)
            p[0].record_stream(cur)
        _lib.call("pf_wgrad_reduce_batch_f32", parts, dws, elems, splits, rows, taps, swapped, n, 1, _lib.stream(),
                  algo_bytes=4.0 * sum(p[2] * p[3] for p in pending))


def _with_packs(backward):
    """A node's backward runs under the packs its forward ran under (loss.backward() is called outside the context)."""
    def wrapped(ctx, *grads):
        with use_packs(getattr(ctx, "packs", None)):
            try:
                return backward(ctx, *grads)
            finally:
                _reduce_flush()
    return wrapped


# ---------------------------------------------------------------------------------------------
# thin wrappers over the C ABI
# ---------------------------------------------------------------------------------------------
def bn_train_rows(bn, partials, col0, C, count, G, groups_per_stat, unbias_n=None, rows=None, col_out=0, ch0=0,
                  bump=True):
    """(4, S, ld) rows [scale | shift | mean | invstd] of a train-mode BatchNorm from the statistics partials
    (G, T, pcols, 2) of its producer; updates the running statistics like the module would (one update per group).
    ``rows`` / ``col_out`` / ``ch0``: write channels [ch0, ch0 + C) of the module into columns [col_out, col_out + C)
    of an existing rows tensor (EdgeConv's BatchNorm: central and difference halves have separate statistics)."""
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm momentum is not supported")
    S = G // groups_per_stat
    if rows is None:
        rows = torch.empty((4, S, C), dtype=_F32, device=partials.device)
    track = bn.track_running_stats and bn.running_mean is not None
    _lib.call("pf_bn_train_rows_f32", _lib.ptr(partials), int(partials.shape[1]), int(partials.shape[2]), int(col0),
              int(C), float(count), float(count if unbias_n is None else unbias_n), _lib.ptr(bn.weight.detach()),
              _lib.ptr(bn.bias.detach()), _lib.ptr(bn.running_mean if track else None),
              _lib.ptr(bn.running_var if track else None), float(bn.momentum), float(bn.eps), int(G),
              int(groups_per_stat), _lib.ptr(rows), int(rows.shape[2]), int(col_out), int(ch0), _lib.stream(),
              algo_bytes=16.0 * partials.shape[0] * partials.shape[1] * C)
    if bump:
        pointflow.bump_counter(bn, S)
    return rows


# PF_TRAIN_LAZY_BN=1 (default since round 6: measured on the driver's box in BENCH_r05 -- the step's loss, gradient and
# running statistics equal the eager-finalize step's bit for bit, +1.7 % train-scenes/s): the training forward finalizes a
# BatchNorm on the critical path only where the rows are needed at once (a materialised activation, EdgeConv's apply
# pass).  Everywhere else the CONSUMER resolves the pending BatchNorm from the producer's statistics rows (``in_bn``,
# csrc/pf_bn_resolve.h, as the inference path does), and the rows tensor (4, S, C) the BACKWARD reads -- [scale | shift |
# mean | invstd] -- is written for all such layers by the batched finalize at the end of the forward
# (pointflow.flush_lazy_stats: <= 32 jobs per launch, MAX_BN_JOBS), together with the running statistics.
# 0: one pf_bn_train_rows_f32 launch per BatchNorm (round 4: 46 launches of ~6.8 us in the chain).
TRAIN_LAZY_BN = int(os.environ.get("PF_TRAIN_LAZY_BN", "1"))


class _Rows(object):
    """The rows tensor of a train-mode BatchNorm and its pending finalize: ``pending()`` is what the next kernel takes
    as its input affine (a pointflow.LazyAffine while the finalize has not run, else the (scale, shift) rows),
    ``now()`` runs the finalize if it has not run and returns the rows tensor."""

    def __init__(self, rows, lazy):
        self.rows, self.lazy = rows, lazy

    def pending(self):
        return (self.rows[0], self.rows[1]) if (self.lazy is None or self.lazy.done) else self.lazy

    def now(self):
        if self.lazy is not None and not self.lazy.done:
            self.lazy.rows()
        return self.rows


def bn_rows(bn, partials, C, count, G, groups_per_stat, track=True, lazy=True):
    """(4, S, C) rows of a train-mode BatchNorm from its producer's statistics partials (G, T, C, 2) as a _Rows: one
    pf_bn_job that writes all four rows and (``track``) the running statistics -- deferred to the batched finalize at the
    end of the forward when ``lazy`` and the consumer-side resolve is affordable (pointflow.LAZY_MAX_ELEMS)."""
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm momentum is not supported")
    S = G // groups_per_stat
    rows = torch.empty((4, S, C), dtype=_F32, device=partials.device)
    if not TRAIN_LAZY_BN:
        if not track:                                       # (before anything is launched: no running statistic moves)
            raise RuntimeError("bn_rows(track=False) needs PF_TRAIN_LAZY_BN=1")
        bn_train_rows(bn, partials, 0, C, count, G, groups_per_stat, rows=rows, bump=True)
        return _Rows(rows, None)
    job = pointflow.bn_job(bn, partials, 0, C, count, count, G, groups_per_stat, rows[0], rows[1])
    job.rows4 = 1                                           # rows[2], rows[3]: mean, invstd
    if not track:                                           # (the caller's normalise pass updates them)
        job.running_mean = job.running_var = None
    else:
        pointflow.bump_counter(bn, S)
    z = pointflow.LazyAffine(job, (partials, rows) + pointflow._bn_tensors(bn), rows[0], rows[1])
    z.origin = torch.cuda.current_stream(partials.device)
    if not track:
        z.defer()                                           # nobody resolves it: only the backward reads these rows
    elif not (lazy and pointflow.LAZY_BN and groups_per_stat * partials.shape[1] * C <= pointflow.LAZY_MAX_ELEMS):
        z.rows()                                            # too many statistics rows for a consumer block: finalize now
    else:
        z.defer()
    return _Rows(rows, z)


def channel_bn_apply(y, bn, partials, addend=None):
    """z = relu(BatchNorm(y)) (+ addend), out of place, with the finalize inside the pass (pf_channel_bn_apply_f32: every
    block reduces its (group, channel) partials itself) and the running statistics updated by it: VolumeConv's materialised
    activations and its decoder's skip adds (reference networks.py:163-166) without a finalize launch in front."""
    N, C = y.shape[:2]
    S = y[0, 0].numel()
    z = torch.empty_like(y)
    track = bn.track_running_stats and bn.running_mean is not None
    _lib.call("pf_channel_bn_apply_f32", _lib.ptr(y), _lib.ptr(z), _lib.ptr(partials), int(partials.shape[1]), N, C, S, 1,
              float(S), _lib.ptr(bn.weight.detach()), _lib.ptr(bn.bias.detach()),
              _lib.ptr(bn.running_mean if track else None), _lib.ptr(bn.running_var if track else None),
              float(bn.momentum), float(bn.eps), 1, _lib.ptr(addend), _lib.stream(),
              algo_bytes=(8.0 if addend is None else 12.0) * N * C * S)
    pointflow.bump_counter(bn, N)
    return z


def channel_affine(y, rows, samples_per_stat, relu=True):
    """z = act(y * scale + shift), out of place (y stays: the backward needs the raw convolution output)."""
    N, C = y.shape[:2]
    S = y[0, 0].numel()
    z = torch.empty_like(y)
    _lib.call("pf_channel_affine_f32", _lib.ptr(y), _lib.ptr(z), _lib.ptr(rows[0]), _lib.ptr(rows[1]), N, C, S,
              int(samples_per_stat), int(bool(relu)), _lib.stream(), algo_bytes=8.0 * N * C * S)
    return z


BN_BWD_PLANE = int(os.environ.get("PF_BN_BWD_PLANE", "1"))      # 0: always the two-launch form


def bn_backward(g, y, rows, samples_per_stat, relu=True, into=None):
    """BatchNorm(+ReLU) backward on planar (N, C, *spatial) tensors: (dy, dgamma, dbeta).  ``into`` = (dgamma, dbeta)
    tensors to ADD the parameter gradients to (then the returned ones are None)."""
    N, C = y.shape[:2]
    S = y[0, 0].numel()
    g = g.contiguous()
    dev = y.device
    # (one block per channel walks the N samples: with three views of 20 480 elements on 32 blocks -- the towers' 32-channel
    # layers -- that is 21 us against 13.5 for the two launches, r06_cfg4_last_steps.md; so: one sample, or small planes)
    if (BN_BWD_PLANE and y.is_contiguous() and (N == 1 or S <= 8192)
            and _lib.load().pf_bn_bwd_plane_supported(S, int(samples_per_stat))):
        # a plane that fits one block's registers: reduce + coefficients + apply in ONE launch (csrc/norm_bwd.hip)
        if into is None:
            dgamma = torch.empty((C,), dtype=_F32, device=dev)
            dbeta = torch.empty((C,), dtype=_F32, device=dev)
        else:
            dgamma, dbeta = into
        dy = torch.empty_like(y)
        _lib.call("pf_bn_bwd_plane_f32", _lib.ptr(g), _lib.ptr(y), _lib.ptr(rows), N, C, S, int(bool(relu)), _lib.ptr(dy),
                  _lib.ptr(dgamma), _lib.ptr(dbeta), 0 if into is None else 1, _lib.stream(), algo_bytes=12.0 * N * C * S)
        return (dy, dgamma, dbeta) if into is None else (dy, None, None)
    T = int(_lib.load().pf_norm_blocks(S))
    partials = torch.empty((N, T, C, 2), dtype=torch.float64, device=dev)
    _lib.call("pf_bn_bwd_reduce_f32", _lib.ptr(g), _lib.ptr(y), _lib.ptr(rows), N, C, S, int(samples_per_stat),
              int(bool(relu)), _lib.ptr(partials), _lib.stream(), algo_bytes=8.0 * N * C * S)
    if into is None:
        dgamma = torch.empty((C,), dtype=_F32, device=dev)
        dbeta = torch.empty((C,), dtype=_F32, device=dev)
    else:
        dgamma, dbeta = into
    dy = torch.empty_like(y)
    # the coefficients ride in the apply pass: every block re-adds its group's few dozen partial rows (one launch less
    # per BatchNorm in a chain of ~600)
    _lib.call("pf_bn_bwd_apply_fused_f32", _lib.ptr(g), _lib.ptr(y), _lib.ptr(rows), _lib.ptr(partials), T,
              float(samples_per_stat) * S, _lib.ptr(dy), N, C, S, int(samples_per_stat), int(bool(relu)), _lib.ptr(dgamma),
              _lib.ptr(dbeta), 0 if into is None else 1, _lib.stream(), algo_bytes=12.0 * N * C * S)
    return (dy, dgamma, dbeta) if into is None else (dy, None, None)


def rows_bn_backward(g, y, rows, C, G, Ng, groups_per_stat, relu=True, into=None):
    """The same on point-major rows: g, y (G*Ng, ld) views whose first C columns are used."""
    dev = y.device
    T = int(_lib.load().pf_rows_bn_blocks(int(G), int(Ng)))
    partials = torch.empty((G, T, C, 2), dtype=torch.float64, device=dev)
    _lib.call("pf_rows_bn_bwd_reduce_f32", _lib.ptr(g), int(g.stride(0)), _lib.ptr(y), int(y.stride(0)), _lib.ptr(rows),
              int(C), int(G), int(Ng), int(groups_per_stat), int(bool(relu)), _lib.ptr(partials), _lib.stream(),
              algo_bytes=8.0 * G * Ng * C)
    S = G // groups_per_stat
    coef = torch.empty((2, S, C), dtype=_F32, device=dev)
    if into is None:
        dgamma = torch.empty((C,), dtype=_F32, device=dev)
        dbeta = torch.empty((C,), dtype=_F32, device=dev)
    else:
        dgamma, dbeta = into
    _lib.call("pf_bn_bwd_coeffs_f32", _lib.ptr(partials), T, C, 0, C, float(groups_per_stat) * Ng, int(G),
              int(groups_per_stat), _lib.ptr(rows), _lib.ptr(coef), _lib.ptr(dgamma), _lib.ptr(dbeta),
              0 if into is None else 1, _lib.stream(), algo_bytes=16.0 * G * T * C)
    dy = torch.empty((G * Ng, C), dtype=_F32, device=dev)
    _lib.call("pf_rows_bn_bwd_apply_f32", _lib.ptr(g), int(g.stride(0)), _lib.ptr(y), int(y.stride(0)), _lib.ptr(rows),
              _lib.ptr(coef), _lib.ptr(dy), C, int(C), int(G), int(Ng), int(groups_per_stat), int(bool(relu)),
              _lib.stream(), algo_bytes=12.0 * G * Ng * C)
    return (dy, dgamma, dbeta) if into is None else (dy, None, None)


def rows_affine(y, rows, C, G, Ng, groups_per_stat, relu=True):
    z = torch.empty((G * Ng, C), dtype=_F32, device=y.device)
    _lib.call("pf_rows_affine_f32", _lib.ptr(y), int(y.stride(0)), _lib.ptr(rows), _lib.ptr(z), C, int(C), int(G),
              int(Ng), int(groups_per_stat), int(bool(relu)), _lib.stream(), algo_bytes=8.0 * G * Ng * C)
    return z


def conv_wgrad(gr, x, kernel, stride, pad, x_affine=None, x_samples_per_stat=1, into=None):
    """dw (Cg, Cx, *kernel) = sum gr[n, cg, o] * act(x)[n, cx, o * stride + k - pad] (pf_conv_wgrad_f32).
    gr (N, Cg, *coarse grid), x (N, Cx, *fine grid), 2-D or 3-D; x_affine = (scale, shift) rows of x's pending
    BatchNorm + ReLU or None."""
    nd = gr.dim() - 2
    N, Cg = gr.shape[:2]
    Cx = x.shape[1]
    if (WGRAD_SWAP and stride == 1 and x_affine is None and Cg <= 8 and Cg < Cx and gr.shape[2:] == x.shape[2:]
            and all(2 * p + 1 == k for p, k in zip(pad, kernel))):
        return _conv_wgrad_swapped(gr, x, kernel, pad, into)
    go = (1,) * (3 - nd) + tuple(gr.shape[2:])
    xi = (1,) * (3 - nd) + tuple(x.shape[2:])
    k3 = (1,) * (3 - nd) + tuple(kernel)
    p3 = (0,) * (3 - nd) + tuple(pad)
    lib = _lib.load()
    nbytes = int(lib.pf_conv_wgrad_workspace(N, Cg, Cx, go[0], go[1], go[2], xi[0], xi[1], xi[2], k3[0], k3[1], k3[2],
                                             int(stride)))
    if nbytes < 0:
        raise RuntimeError("conv_wgrad: unsupported shape")
    sc, sh = (None, None) if x_affine is None else x_affine
    taps = k3[0] * k3[1] * k3[2]

    def launch():
        work = torch.empty((max(nbytes, 4) // 4,), dtype=_F32, device=gr.device)
        dw = torch.empty((Cg, Cx) + tuple(kernel), dtype=_F32, device=gr.device) if into is None else into   # into: dw +=
        batched = into is not None and WGRAD_BATCH and DIRECT_GRADS and into.is_contiguous()
        if batched and WGRAD_DEFER:
            _defer_wgrad(gr, x, N, Cg, Cx, go, xi, k3, stride, p3, sc, sh, x_samples_per_stat, work, nbytes,
                         2.0 * N * go[0] * go[1] * go[2] * taps * Cg * Cx, 4.0 * (gr.numel() + x.numel()) + 4.0 * dw.numel())
            _queue_reduce(work, into, Cg * Cx * taps, nbytes, Cg, taps)
            return None
        _lib.call("pf_conv_wgrad_f32", _lib.ptr(gr), _lib.ptr(x), None if batched else _lib.ptr(dw), N, Cg, Cx, go[0],
                  go[1], go[2], xi[0], xi[1], xi[2], k3[0], k3[1], k3[2], int(stride), p3[0], p3[1], p3[2], _lib.ptr(sc),
                  _lib.ptr(sh), int(x_samples_per_stat), _lib.ptr(work), nbytes, 0 if into is None else 1, _lib.stream(),
                  algo_bytes=4.0 * (gr.numel() + x.numel()) + 4.0 * dw.numel(),
                  flops=2.0 * N * go[0] * go[1] * go[2] * taps * Cg * Cx)
        if batched:
            _queue_reduce(work, into, Cg * Cx * taps, nbytes, Cg, taps)
        return dw if into is None else None

    return launch()


# Stride-1 'same' layers with at most 8 output channels (VolumeConv's conv0_1 64 -> 8 and conv6_2 8 -> 1): the operands
# change places (include/pointflow_hip.h, pf_wgrad_reduce_batch_f32's `swapped`) -- the MFMA rows are the INPUT channels
# (64 of 64 rows used instead of 8 of 16) and the patch that is staged with its halo is the 8-channel gradient instead of
# the 64-channel activation.  PF_WGRAD_SWAP=0: the plain form.
WGRAD_SWAP = int(os.environ.get("PF_WGRAD_SWAP", "1"))


def _conv_wgrad_swapped(gr, x, kernel, pad, into):
    """conv_wgrad(gr, x) as pf_conv_wgrad_f32(gr' = x, x' = gr): partials (Cx, Cg, taps) with reversed taps, put into
    nn.ConvNd's order by the batched reduce (the step) or by a transpose + flip (a stand-alone call)."""
    nd = gr.dim() - 2
    N, Cg = gr.shape[:2]
    Cx = x.shape[1]
    sp = (1,) * (3 - nd) + tuple(gr.shape[2:])
    k3 = (1,) * (3 - nd) + tuple(kernel)
    p3 = (0,) * (3 - nd) + tuple(pad)
    taps = k3[0] * k3[1] * k3[2]
    lib = _lib.load()
    nbytes = int(lib.pf_conv_wgrad_workspace(N, Cx, Cg, sp[0], sp[1], sp[2], sp[0], sp[1], sp[2], k3[0], k3[1], k3[2], 1))
    if nbytes < 0:
        raise RuntimeError("conv_wgrad: unsupported shape")
    work = torch.empty((max(nbytes, 4) // 4,), dtype=_F32, device=gr.device)
    batched = into is not None and WGRAD_BATCH and DIRECT_GRADS and into.is_contiguous()
    if batched and WGRAD_DEFER:
        _defer_wgrad(x, gr, N, Cx, Cg, sp, sp, k3, 1, p3, None, None, 1, work, nbytes,
                     2.0 * N * sp[0] * sp[1] * sp[2] * taps * Cg * Cx, 4.0 * (gr.numel() + x.numel()) + 4.0 * Cg * Cx * taps)
        _queue_reduce(work, into, Cg * Cx * taps, nbytes, Cx, taps, swapped=True)
        return None
    dwt = None if batched else torch.empty((Cx, Cg) + tuple(kernel), dtype=_F32, device=gr.device)
    _lib.call("pf_conv_wgrad_f32", _lib.ptr(x), _lib.ptr(gr), _lib.ptr(dwt), N, Cx, Cg, sp[0], sp[1], sp[2], sp[0], sp[1],
              sp[2], k3[0], k3[1], k3[2], 1, p3[0], p3[1], p3[2], None, None, 1, _lib.ptr(work), nbytes, 0, _lib.stream(),
              algo_bytes=4.0 * (gr.numel() + x.numel()) + 4.0 * Cg * Cx * taps,
              flops=2.0 * N * sp[0] * sp[1] * sp[2] * taps * Cg * Cx)
    if batched:
        _queue_reduce(work, into, Cg * Cx * taps, nbytes, Cx, taps, swapped=True)
        return None
    dw = dwt.transpose(0, 1).flip(*range(2, 2 + nd)).contiguous()
    if into is None:
        return dw
    into.add_(dw)
    return None


def rows_wgrad(gr, x, Cg, Cx, x_affine=None, x_rows_per_stat=None, into=None, side=None):
    """dw (Cg, Cx) = sum_p gr[p, :Cg]^T act(x[p, :Cx]) on point-major row views (pf_rows_wgrad_f32).  ``side``: a stream to
    issue the launch (and, later, its reduction) on when the result is ADDED into the gradient bucket -- nothing inside the
    step reads it, so the chain does not have to wait for it (model.TRAIN_FORK = 3)."""
    P = gr.shape[0]
    lib = _lib.load()
    nbytes = int(lib.pf_rows_wgrad_workspace(P, int(Cg), int(Cx)))
    if nbytes < 0:
        raise RuntimeError("rows_wgrad: unsupported shape")
    sc, sh = (None, None) if x_affine is None else x_affine

    def launch():
        work = torch.empty((max(nbytes, 4) // 4,), dtype=_F32, device=gr.device)
        dw = torch.empty((Cg, Cx), dtype=_F32, device=gr.device) if into is None else into
        batched = into is not None and WGRAD_BATCH and DIRECT_GRADS and into.is_contiguous()
        if batched and WGRAD_DEFER and WGRAD_LATE and side is None:
            _LATE["wgrad"].append(dict(gr=gr, x=x, Cg=int(Cg), Cx=int(Cx), sc=sc, sh=sh, work=work, nbytes=int(nbytes),
                                       flops=2.0 * P * Cg * Cx, bytes=4.0 * P * (Cg + Cx) + 4.0 * Cg * Cx,
                                       rows=(int(P), int(gr.stride(0)), int(x.stride(0)), int(x_rows_per_stat or P))))
            _LATE["reduce"].append((work, into, int(Cg) * int(Cx), int(nbytes // (4 * int(Cg) * int(Cx))), int(Cg), 1, 0,
                                    torch.cuda.current_stream(work.device)))
            return None
        _lib.call("pf_rows_wgrad_f32", _lib.ptr(gr), int(gr.stride(0)), _lib.ptr(x), int(x.stride(0)),
                  None if batched else _lib.ptr(dw), P, int(Cg), int(Cx), _lib.ptr(sc), _lib.ptr(sh),
                  int(x_rows_per_stat or P), _lib.ptr(work), nbytes, 0 if into is None else 1, _lib.stream(),
                  algo_bytes=4.0 * P * (Cg + Cx) + 4.0 * Cg * Cx, flops=2.0 * P * Cg * Cx)
        if batched:
            _queue_reduce(work, into, int(Cg) * int(Cx), nbytes, int(Cg), 1)
        return dw if into is None else None

    if side is not None and into is not None and WGRAD_BATCH and DIRECT_GRADS and into.is_contiguous():
        side.wait_stream(torch.cuda.current_stream(gr.device))
        for t in (gr, x) + (() if x_affine is None else tuple(x_affine)):
            t.record_stream(side)
        with torch.cuda.stream(side):
            return launch()
    return launch()


def gemm_rows(x, w, K, n_out, chunks=None):
    """(P, n_out) = x[:, :K] @ w (K, n_out) on point-major rows through pf_pointwise_gemm_f32 (column chunks of at most
    128, zero padded to the kernel's 32 / 64 / 128 widths; ``chunks``: the step's prepacked [(col0, width, wt)])."""
    P = x.shape[0]
    y = torch.empty((P, n_out), dtype=_F32, device=x.device)
    if chunks is None:
        chunks, col = [], 0
        while col < n_out:
            width = min(128, n_out - col)
            nc = 32 if width <= 32 else (64 if width <= 64 else 128)
            wt = torch.zeros((K, nc), dtype=_F32, device=x.device)
            wt[:, :width] = w[:, col:col + width]
            chunks.append((col, width, wt))
            col += width
    for col, width, wt in chunks:
        pointflow.pointwise_gemm(x, True, int(x.stride(0)), wt, y[:, col:], n_out, 1, P, K, width)
    return y


def _flip_t(w):
    """(Cout, Cin, k...) -> (Cin, Cout, k...) with every spatial axis reversed: the weight of the data gradient of a
    stride-1 'same' convolution, itself a convolution."""
    return w.detach().transpose(0, 1).flip(*range(2, w.dim())).contiguous()


def conv2d_dgrad(dy, weight, stride):
    """dL/dx of y = conv2d(x, weight, stride, pad = k // 2) for ImageConv's shapes (3x3 / 1, 5x5 / 2)."""
    N, Cout, Ho, Wo = dy.shape
    Cin, k = weight.shape[1], weight.shape[2]
    dy = dy.contiguous()
    if stride == 1:
        wp = _packed("c2w_dg", weight)
        if wp is None:
            wp = pointflow._pack_conv2d_wide(_flip_t(weight))
        dx = torch.empty((N, Cin, Ho, Wo), dtype=_F32, device=dy.device)
        _lib.call("pf_conv2d_wide_f32", _lib.ptr(dy), _lib.ptr(wp), _lib.ptr(dx), N, Cout, Cin, Ho, Wo, int(k), 1,
                  None, None, None, 1, None, 0, _lib.stream(),
                  algo_bytes=4.0 * N * (Cin + Cout) * Ho * Wo, flops=2.0 * N * Ho * Wo * k * k * Cin * Cout)
        return dx
    wp = _packed("d2_dg", weight)
    if wp is None:
        ncp = (Cin + 15) // 16 * 16
        wp = torch.zeros((Cout // 4, k * k, 4, ncp), dtype=_F32, device=dy.device)
        wp[..., :Cin] = weight.detach().reshape(Cout // 4, 4, Cin, k * k).permute(0, 3, 1, 2)
    dx = torch.empty((N, Cin, 2 * Ho, 2 * Wo), dtype=_F32, device=dy.device)
    _lib.call("pf_deconv2d_k5s2_f32", _lib.ptr(dy), _lib.ptr(wp), _lib.ptr(dx), N, Cout, Cin, Ho, Wo, _lib.stream(),
              algo_bytes=4.0 * N * (Cout + 4 * Cin) * Ho * Wo, flops=2.0 * N * Ho * Wo * k * k * Cin * Cout)
    return dx


def _conv3d_k3_packed(x, wp, Cout, stride, y=None):
    """pf_conv3d_k3_f32 on an already packed weight (Cin/4, 27, 4, 16 ceil(Cout/16)); ``y``: where to write (one sample:
    a channel slice of a wider tensor)."""
    N, Cin, Di, Hi, Wi = x.shape
    Do, Ho, Wo = (Di - 1) // stride + 1, (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    if y is None:
        y = torch.empty((N, Cout, Do, Ho, Wo), dtype=_F32, device=x.device)
    _lib.call("pf_conv3d_k3_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, Cout, Di, Hi, Wi, int(stride), None,
              None, None, 1, None, _lib.stream(), algo_bytes=4.0 * N * (Cin * Di * Hi * Wi + Cout * Do * Ho * Wo),
              flops=2.0 * N * Do * Ho * Wo * 27 * Cin * Cout)
    return y


def conv3d_dgrad_flip(dy, weight):
    """dL/dx of a stride-1 3x3x3 'same' convolution y = conv3d(x, weight): the forward kernel on the flipped, transposed
    weight; Cin <= 32 in one launch, Cin = 64 as two halves of the output channels (one sample)."""
    Cout, Cin = weight.shape[:2]
    if Cin <= 32:
        wp = _packed("c3_dg", weight)
        if wp is None:
            return pointflow.conv3d_k3(dy, _flip_t(weight), 1, False)[0]
        return _conv3d_k3_packed(dy, wp, Cin, 1)
    N, _, D, H, W = dy.shape
    if N != 1 or Cin % 32:
        raise RuntimeError("conv3d dgrad: unsupported shape")
    dx = torch.empty((1, Cin, D, H, W), dtype=_F32, device=dy.device)
    wf = None
    for h, c0 in enumerate(range(0, Cin, 32)):
        wp = _packed("c3_dg%d" % h, weight)
        if wp is None:
            wf = _flip_t(weight) if wf is None else wf
            wp = pointflow.pack_conv3d_weight(wf[c0:c0 + 32].contiguous())
        _conv3d_k3_packed(dy, wp, 32, 1, y=dx[:, c0:])
    return dx


def _conv3d_k3_w(x, w, stride):
    """conv3d 3x3x3 / pad 1 with the weight tensor (Cout <= 32, Cin, 3, 3, 3) read as it is (pf_conv3d_k3_f32; a
    ConvTranspose3d weight read this way gives the transposed layer's data gradient at stride 2)."""
    return pointflow.conv3d_k3(x, w, stride, False)[0]


def _conv3d_bottom_w(x, w, stride, flip_t=False):
    """pf_conv3d_bottom_f32 (Cout = 64; 32 -> 64 / 2 or 64 -> 64 / 1) with the weight (64, Cin, 3, 3, 3) as it is, or
    flipped and transposed (``flip_t``: the stride-1 layer's data gradient)."""
    N, Cin, Di, Hi, Wi = x.shape
    Do, Ho, Wo = (Di - 1) // stride + 1, (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    wp = _packed("c3b_dg" if flip_t else "c3b", w)
    if wp is None:
        wt = _flip_t(w) if flip_t else w.detach().to(_F32)
        wp = wt.permute(2, 3, 4, 1, 0).reshape(3, 3, 3, Cin // 16, 4, 4, 64).permute(0, 1, 2, 3, 4, 6, 5).contiguous()
    y = torch.empty((N, 64, Do, Ho, Wo), dtype=_F32, device=x.device)
    _lib.call("pf_conv3d_bottom_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, 64, Di, Hi, Wi, int(stride), None,
              None, None, 1, None, _lib.stream(), algo_bytes=4.0 * N * (Cin * Di * Hi * Wi + 64 * Do * Ho * Wo),
              flops=2.0 * N * Do * Ho * Wo * 27 * Cin * 64)
    return y


def _deconv3d_bottom_w(x, w):
    """pf_deconv3d_bottom_f32 with the weight (64, 32, 3, 3, 3) read in ConvTranspose3d's layout."""
    N, Cin, Di, Hi, Wi = x.shape
    wp = _packed("d3b", w)
    if wp is None:
        wp = w.detach().to(_F32).permute(2, 3, 4, 0, 1).reshape(27, Cin // 16, 4, 4, 32).permute(0, 1, 2, 4, 3).contiguous()
    y = torch.empty((N, 32, 2 * Di, 2 * Hi, 2 * Wi), dtype=_F32, device=x.device)
    _lib.call("pf_deconv3d_bottom_f32", _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), N, Cin, 32, Di, Hi, Wi, None, None, None,
              1, None, _lib.stream(), algo_bytes=4.0 * N * Di * Hi * Wi * (Cin + 8 * 32),
              flops=2.0 * N * Di * Hi * Wi * 27 * Cin * 32)
    return y


# ---------------------------------------------------------------------------------------------
# ImageConv tower (reference networks.py:84-124), all V views of one scene per launch, per-view BatchNorm statistics
# ---------------------------------------------------------------------------------------------
_STAGES = ("conv0", "conv1", "conv2", "conv3")


def _tower_blocks(tower):
    out = []
    for name in _STAGES:
        seq = getattr(tower, name)
        for i, blk in enumerate(seq):
            conv, bn = (blk.conv, blk.bn) if hasattr(blk, "bn") else (blk, None)
            out.append((name, i + 1 == len(seq), conv, bn))
    return out


def tower_supported(tower, img):
    """The fused training path of a tower: every layer a tower-kernel shape with a train-mode BatchNorm + ReLU (the
    last one plain), stride-2 data gradients within the transposed kernel's widths."""
    if img.dim() != 4 or img.dtype != _F32 or not img.is_cuda:
        return False
    # three stride-2 stages: the data gradient of a 5x5 / 2 layer (pf_deconv2d_k5s2_f32) writes (2 Ho, 2 Wo), which is
    # the layer's input size only when that is even at every stage; other sizes take the composed ATen path
    if img.shape[-2] % 8 or img.shape[-1] % 8:
        return False
    blocks = _tower_blocks(tower)
    for i, (name, last, conv, bn) in enumerate(blocks):
        if not pointflow.conv2d_wide_supported(conv):
            return False
        blk = getattr(tower, name)
        if bn is not None and not (bn.training and bn.affine and bn.momentum is not None):
            return False
        if conv.stride[0] == 2 and not _lib.load().pf_deconv2d_k5s2_supported(conv.out_channels, conv.in_channels):
            return False
        if conv.stride[0] == 1 and i > 0 and not _lib.load().pf_conv2d_wide_supported(
                conv.out_channels, conv.in_channels, int(conv.kernel_size[0]), 1):
            return False
    for name in _STAGES:
        for blk in getattr(tower, name):
            if hasattr(blk, "bn") and not blk.relu:
                return False
    return blocks[-1][3] is None and all(b[3] is not None for b in blocks[:-1])


class _TowerTrain(torch.autograd.Function):
    """(V, 3, H, W) -> the stage outputs named in ``want`` (normalised; "conv3" is the plain last convolution)."""

    @staticmethod
    def forward(ctx, img, tower, want, *params):
        ctx.packs = _PACKS
        blocks = _tower_blocks(tower)
        V = img.shape[0]
        x = img.detach().contiguous()
        saved, outs = [], []
        pending = None
        with torch.cuda.device(x.device):
            prev = None                                      # the previous layer's _Rows (its BatchNorm + ReLU is pending)
            for name, stage_end, conv, bn in blocks:
                y, partials = pointflow.conv2d_wide(x, conv, None if prev is None else prev.pending(), 1, bn is not None)
                cur = None
                if bn is not None:
                    cur = bn_rows(bn, partials, conv.out_channels, float(y[0, 0].numel()), V, 1)
                saved.append((x, None if prev is None else (prev.rows[0], prev.rows[1]), y,
                              None if cur is None else cur.rows))
                if stage_end and name in want:
                    outs.append(y if cur is None else channel_affine(y, cur.now(), 1, True))
                x, prev = y, cur
            pointflow.flush_counters()                       # (the deferred finalizes: here, or at the end of the step's forward)
        ctx.tower, ctx.want, ctx.saved = tower, tuple(want), saved
        return tuple(outs)

    @staticmethod
    @_with_packs
    def backward(ctx, *grads):
        tower, want, saved = ctx.tower, ctx.want, ctx.saved
        blocks = _tower_blocks(tower)
        incoming = dict(zip([n for n in _STAGES if n in want], grads))
        gparams = [None] * sum(1 if b[3] is None else 3 for b in blocks)
        slots, k = [], 0
        for b in blocks:
            slots.append(k)
            k += 1 if b[3] is None else 3
        g = None
        with torch.cuda.device(saved[0][0].device):
            for i in range(len(blocks) - 1, -1, -1):
                name, stage_end, conv, bn = blocks[i]
                x, pending, y, rows = saved[i]
                if stage_end and incoming.get(name) is not None:
                    gi = incoming[name].contiguous()
                    g = gi if g is None else g + gi
                if g is None:
                    continue
                if g.shape != y.shape:
                    raise RuntimeError("tower backward: gradient %s for an activation %s" % (tuple(g.shape), tuple(y.shape)))
                if bn is not None:
                    tg, tb = _grad_target(bn.weight), _grad_target(bn.bias)
                    into = (tg, tb) if (tg is not None and tb is not None) else None
                    dy, dgamma, dbeta = bn_backward(g, y, rows, 1, True, into=into)
                    gparams[slots[i] + 1], gparams[slots[i] + 2] = dgamma, dbeta
                else:
                    dy = g.contiguous()
                ks, st = int(conv.kernel_size[0]), int(conv.stride[0])
                gparams[slots[i]] = conv_wgrad(dy, x, (ks, ks), st, (ks // 2, ks // 2), x_affine=pending,
                                               into=_grad_target(conv.weight))
                g = conv2d_dgrad(dy, conv.weight, st) if i > 0 else None
        return (None, None, None) + tuple(gparams)


def tower_params(tower):
    ps = []
    for _, _, conv, bn in _tower_blocks(tower):
        ps.append(conv.weight)
        if bn is not None:
            ps += [bn.weight, bn.bias]
    return ps


def tower_train(tower, img, want):
    """Stage outputs {name: (V, c, h, w)} of ``tower`` for the views ``img`` (V, 3, H, W) with the hand-written
    backward attached."""
    want = tuple(n for n in _STAGES if n in want)
    outs = _TowerTrain.apply(img, tower, want, *tower_params(tower))
    return dict(zip(want, outs))


# ---------------------------------------------------------------------------------------------
# VolumeConv (reference networks.py:127-167), one scene
# ---------------------------------------------------------------------------------------------
def volume_supported(vc, x):
    if x.dim() != 5 or x.shape[0] != 1 or x.dtype != _F32 or not x.is_cuda:
        return False
    if vc.in_channels != 64 or vc.base_channels != 8 or not vc._bottom_fusable():
        return False
    D, H, W = x.shape[2:]
    if D % 8 or H % 8 or W % 8:
        return False
    for name in ("conv0_1", "conv1_0", "conv2_0", "conv3_0", "conv1_1", "conv2_1", "conv3_1", "conv4_0", "conv5_0",
                 "conv6_0"):
        blk = getattr(vc, name)
        if blk.bn is None or not blk.relu or not (blk.bn.training and blk.bn.affine and blk.bn.momentum is not None):
            return False
    return type(vc.conv6_2) is torch.nn.Conv3d and vc.conv6_2.bias is None


_VC_BLOCKS = ("conv0_1", "conv1_0", "conv2_0", "conv3_0", "conv3_1", "conv1_1", "conv2_1", "conv4_0", "conv5_0",
              "conv6_0")


def volume_params(vc):
    ps = []
    for name in _VC_BLOCKS:
        blk = getattr(vc, name)
        ps += [blk.conv.weight, blk.bn.weight, blk.bn.bias]
    ps.append(vc.conv6_2.weight)
    return ps


class _VolumeTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, vc, *params):
        ctx.packs = _PACKS
        x0 = cost.detach().contiguous()
        # one scene per process (volume_supported): both BatchNorm forms below advance num_batches_tracked by ONE here
        # (the normalise pass counts samples, the rows launch statistic groups; they agree only for a batch of one)
        assert x0.shape[0] == 1, "the fused VolumeConv training node is built for one scene per call"
        rec = {}

        def bn_act(name, y, partials, addend=None):
            blk = getattr(vc, name)
            S = y[0, 0].numel()
            if not TRAIN_LAZY_BN:
                rows = bn_train_rows(blk.bn, partials, 0, y.shape[1], float(S), 1, 1)
                z = channel_affine(y, rows, 1, True)
                return rows, (z if addend is None else z + addend)
            # the normalise pass finalizes for itself (and adds the decoder's skip tensor); the rows the backward reads come
            # from the batched finalize at the end of the forward
            z = channel_bn_apply(y, blk.bn, partials, addend)
            return bn_rows(blk.bn, partials, y.shape[1], float(S), 1, 1, track=False, lazy=True).rows, z

        def conv(name, x, stride):
            blk = getattr(vc, name)
            y, p = pointflow.conv3d_k3(x, blk.conv.weight, stride, True)
            rows, z = bn_act(name, y, p)
            rec[name] = (x, y, rows)
            return z

        with torch.cuda.device(x0.device):
            z01 = conv("conv0_1", x0, 1)
            z10 = conv("conv1_0", x0, 2)
            z20 = conv("conv2_0", z10, 2)
            y30, p30 = pointflow.conv3d_bottom(z20, vc.conv3_0.conv, None, 1, True)
            r30, z30 = bn_act("conv3_0", y30, p30)
            rec["conv3_0"] = (z20, y30, r30)
            y31, p31 = pointflow.conv3d_bottom(z30, vc.conv3_1.conv, None, 1, True)
            r31, z31 = bn_act("conv3_1", y31, p31)
            rec["conv3_1"] = (z30, y31, r31)
            z11 = conv("conv1_1", z10, 1)
            z21 = conv("conv2_1", z20, 1)
            y40, p40 = pointflow.deconv3d_bottom(z31, vc.conv4_0.conv, None, 1, True)
            r40, s5 = bn_act("conv4_0", y40, p40, z21)                 # s5 = z40 + z21 (networks.py:163)
            rec["conv4_0"] = (z31, y40, r40)
            y50, p50 = pointflow.deconv3d_k3s2(s5, None, vc.conv5_0.conv.weight, True)
            r50, s6 = bn_act("conv5_0", y50, p50, z11)                 # s6 = z50 + z11
            rec["conv5_0"] = (s5, y50, r50)
            y60, p60 = pointflow.deconv3d_k3s2(s6, None, vc.conv6_0.conv.weight, True)
            r60, s7 = bn_act("conv6_0", y60, p60, z01)                 # s7 = z60 + z01
            rec["conv6_0"] = (s6, y60, r60)
            out = pointflow.conv3d_k3_few(s7, vc.conv6_2.weight)
            rec["conv6_2"] = (s7, None, None)
            pointflow.flush_counters()
        ctx.vc, ctx.rec = vc, rec
        return out

    @staticmethod
    @_with_packs
    def backward(ctx, gout):
        vc, rec = ctx.vc, ctx.rec
        grads = {}
        K3, P3 = (3, 3, 3), (1, 1, 1)

        def bnb(name, g):
            _, y, rows = rec[name]
            bn = getattr(vc, name).bn
            tg, tb = _grad_target(bn.weight), _grad_target(bn.bias)
            into = (tg, tb) if (tg is not None and tb is not None) else None
            dy, dgamma, dbeta = bn_backward(g, y, rows, 1, True, into=into)
            grads[name + ".bn"] = (dgamma, dbeta)
            return dy

        def conv_back(name, g, stride):          # Conv3d block: returns dy
            dy = bnb(name, g)
            grads[name] = conv_wgrad(dy, rec[name][0], K3, stride, P3, into=_grad_target(getattr(vc, name).conv.weight))
            return dy

        def deconv_back(name, g):                # Deconv3d block: weight gradient in (Cin, Cout, 3, 3, 3) order
            dy = bnb(name, g)
            grads[name] = conv_wgrad(rec[name][0], dy, K3, 2, P3, into=_grad_target(getattr(vc, name).conv.weight))
            return dy

        with torch.cuda.device(gout.device):
            g = gout.contiguous()
            w62 = vc.conv6_2.weight
            grads["conv6_2"] = conv_wgrad(g, rec["conv6_2"][0], K3, 1, P3, into=_grad_target(w62))
            wf = _packed("c1_dg", w62)
            if wf is None:
                wf = w62.detach().flip(2, 3, 4).reshape(w62.shape[1], 27).contiguous()
            D, H, W = g.shape[2:]
            g7 = torch.empty((1, w62.shape[1], D, H, W), dtype=_F32, device=g.device)
            _lib.call("pf_conv3d_k3_c1_f32", _lib.ptr(g), _lib.ptr(wf), _lib.ptr(g7), 1, int(w62.shape[1]), D, H, W,
                      _lib.stream(), algo_bytes=4.0 * (1 + w62.shape[1]) * D * H * W)
            # decoder: a ConvTranspose3d's data gradient is the stride-2 convolution with its weight read (Cout', Cin')
            dy60 = deconv_back("conv6_0", g7)
            g6 = _conv3d_k3_w(dy60, vc.conv6_0.conv.weight, 2)                        # -> dz50, dz11
            dy50 = deconv_back("conv5_0", g6)
            g5 = _conv3d_k3_w(dy50, vc.conv5_0.conv.weight, 2)                        # -> dz40, dz21
            dy40 = deconv_back("conv4_0", g5)
            g31 = _conv3d_bottom_w(dy40, vc.conv4_0.conv.weight, 2)                   # -> dz31
            dy31 = conv_back("conv3_1", g31, 1)
            g30 = _conv3d_bottom_w(dy31, vc.conv3_1.conv.weight, 1, flip_t=True)
            dy30 = conv_back("conv3_0", g30, 2)
            g20 = _deconv3d_bottom_w(dy30, vc.conv3_0.conv.weight)                    # stride-2 conv: transposed kernel
            dy21 = conv_back("conv2_1", g5, 1)
            g20 = g20 + conv3d_dgrad_flip(dy21, vc.conv2_1.conv.weight)
            dy20 = conv_back("conv2_0", g20, 2)
            g10 = pointflow.deconv3d_k3s2(dy20, None, vc.conv2_0.conv.weight.detach(), False)[0]
            dy11 = conv_back("conv1_1", g6, 1)
            g10 = g10 + conv3d_dgrad_flip(dy11, vc.conv1_1.conv.weight)
            dy10 = conv_back("conv1_0", g10, 2)
            gx = pointflow.deconv3d_k3s2(dy10, None, vc.conv1_0.conv.weight.detach(), False)[0]
            dy01 = conv_back("conv0_1", g7, 1)
            gx = gx + conv3d_dgrad_flip(dy01, vc.conv0_1.conv.weight)
        out = [gx, None]
        for name in _VC_BLOCKS:
            out += [grads[name], grads[name + ".bn"][0], grads[name + ".bn"][1]]
        out.append(grads["conv6_2"])
        return tuple(out)


def volume_train(vc, cost):
    return _VolumeTrain.apply(cost, vc, *volume_params(vc))


# ---------------------------------------------------------------------------------------------
# PointFlow chain: EdgeConv x3 on point-major rows, then the flow MLP (reference model.py:205-216, networks.py:9-81)
# ---------------------------------------------------------------------------------------------
def edge_chain_supported(edge_convs, feature, idx):
    if feature.dim() != 2 or not feature.is_cuda or feature.dtype != _F32 or idx.dim() != 3 or idx.shape[0] != 1:
        return False
    cin = feature.shape[1]
    for m in edge_convs:
        C = m.conv1.weight.shape[0]
        if C not in (32, 64) or m.conv1.weight.shape[1] != cin or cin % 4:
            return False
        if not (m.bn.training and m.bn.affine and m.bn.momentum is not None):
            return False
        cin = (2 if m.concat else 1) * C
    return True


def edge_chain_params(edge_convs):
    ps = []
    for m in edge_convs:
        ps += [m.conv1.weight, m.conv2.weight, m.bn.weight, m.bn.bias]
    return ps


class _EdgeChainTrain(torch.autograd.Function):
    """feature (N, Cin) point-major rows + idx (1, N, k) -> the (N, sum widths) concat buffer of the EdgeConv outputs
    (reference model.py:209-216: each layer's output is the next one's input and a slice of the MLP's input)."""

    @staticmethod
    def forward(ctx, feature, idx, edge_convs, lattice, *params):
        ctx.packs = _PACKS
        ctx.side = _on_side(3)
        x = feature.detach().contiguous()
        N, cin = x.shape
        idx = idx.contiguous()
        widths = [(2 if m.concat else 1) * m.conv1.weight.shape[0] for m in edge_convs]
        ctot = sum(widths)
        edges = torch.empty((N, ctot), dtype=_F32, device=x.device)
        keeps = []
        X, ldx, K, col = x, cin, cin, 0
        with torch.cuda.device(x.device):
            for m, wdt in zip(edge_convs, widths):
                keep = {}
                Y = edges[:, col:]
                pointflow.edge_conv_fused(X, True, ldx, K, 1, N, idx, m.conv1.weight, m.conv2.weight, m.bn, m.concat, Y,
                                          ctot, groups_per_stat=1, lattice=lattice, keep=keep)
                keeps.append((keep, X, ldx, K, col, wdt))
                X, ldx, K = Y, ctot, wdt
                col += wdt
            pointflow.flush_counters()
        ctx.edge_convs, ctx.keeps, ctx.idx, ctx.N, ctx.ctot = edge_convs, keeps, idx, N, ctot
        return edges

    @staticmethod
    @_with_packs
    def backward(ctx, gedges):
        edge_convs, keeps, idx, N = ctx.edge_convs, ctx.keeps, ctx.idx, ctx.N
        k = idx.shape[2]
        g = gedges.contiguous()
        carry = None                               # the next layer's data gradient: added to this layer's column slice
        gparams = []
        gx = None
        with torch.cuda.device(g.device):
            for m, (keep, X, ldx, K, col, wdt) in reversed(list(zip(edge_convs, keeps))):
                C = m.conv1.weight.shape[0]
                gy = g[:, col:col + wdt]
                tg, tb = _grad_target(m.bn.weight), _grad_target(m.bn.bias)
                grad_le, dgamma, dbeta = pointflow.edge_conv_backward(
                    keep, idx, gy, C, k, 1, N, 1, m.concat, into=(tg, tb) if (tg is not None and tb is not None) else None,
                    grad_acc=carry)                      # (+ the next layer's data gradient, added inside the first pass)
                chunks = _packed("rows", m.conv1.weight)
                wcat = None if chunks is not None else torch.cat(
                    [m.conv1.weight.detach().reshape(C, K), m.conv2.weight.detach().reshape(C, K)], dim=0)
                # conv1 / conv2 are adjacent parameters: in the flat gradient bucket their .grad views form one (2C, K) block
                t1, t2 = _grad_target(m.conv1.weight), _grad_target(m.conv2.weight)
                into = None
                if t1 is not None and t2 is not None and t2.data_ptr() == t1.data_ptr() + 4 * C * K:
                    into = torch.as_strided(t1, (2 * C, K), (K, 1))
                dw = rows_wgrad(grad_le, X, 2 * C, K, into=into, side=ctx.side)
                dX = gemm_rows(grad_le, wcat, 2 * C, K, chunks)                            # (N, K)
                if col == 0:
                    gx = dX
                else:
                    carry = dX
                gparams = [None if dw is None else dw[:C].reshape(m.conv1.weight.shape),
                           None if dw is None else dw[C:].reshape(m.conv2.weight.shape), dgamma, dbeta] + gparams
        return (gx, None, None, None) + tuple(gparams)


def edge_chain_train(edge_convs, feature, idx, plane_hw=None):
    """``plane_hw`` = (H, W) of the D x H x W lattice ``idx`` was searched on (model.py:_sub_flow_autograd), or None: a hint
    for the XCD-aware block order of the gather passes (csrc/edgeconv.hip: xcd_tile); results do not depend on it."""
    lattice = None if plane_hw is None else (0, int(plane_hw[0]), int(plane_hw[1]))
    return _EdgeChainTrain.apply(feature, idx, edge_convs, lattice, *edge_chain_params(edge_convs))


def mlp_supported(shared, x):
    if x.dim() != 2 or not x.is_cuda or x.dtype != _F32 or x.shape[1] % 4:
        return False
    cin = x.shape[1]
    for blk in shared:
        conv, bn = blk.conv, blk.bn
        if (type(conv) is not torch.nn.Conv1d or conv.kernel_size != (1,) or conv.bias is not None or bn is None
                or not blk.relu or conv.in_channels != cin or conv.out_channels not in (16, 32, 64, 128)):
            return False
        if not (bn.training and bn.affine and bn.momentum is not None):
            return False
        cin = conv.out_channels
    return len(shared) >= 1


def mlp_params(shared):
    ps = []
    for blk in shared:
        ps += [blk.conv.weight, blk.bn.weight, blk.bn.bias]
    return ps


class _MLPTrain(torch.autograd.Function):
    """SharedMLP over point-major rows (reference nn/mlp.py:45-81): (N, Cin) -> the last block's normalised (N, Cout)."""

    @staticmethod
    def forward(ctx, x, shared, *params):
        ctx.packs = _PACKS
        ctx.side = _on_side(3)
        X = x.detach()
        if X.stride(1) != 1:
            X = X.contiguous()
        N = X.shape[0]
        saved = []
        prev = None
        ldx, K = int(X.stride(0)), X.shape[1]
        with torch.cuda.device(X.device):
            for blk in shared:
                Wt, cout = pointflow.pack_weight_t(blk.conv.weight)
                Z = torch.empty((N, cout), dtype=_F32, device=X.device)
                part = pointflow.pointwise_gemm(X, True, ldx, Wt, Z, cout, 1, N, K, cout,
                                                in_affine=None if prev is None else prev.pending(), want_stats=True)
                cur = bn_rows(blk.bn, part, cout, float(N), 1, 1)
                saved.append((X, None if prev is None else (prev.rows[0], prev.rows[1]), Z, cur.rows, K, cout))
                X, ldx, K, prev = Z, cout, cout, cur
            out = rows_affine(X, prev.now(), K, 1, N, 1, True)
            pointflow.flush_counters()
        ctx.shared, ctx.saved, ctx.N = shared, saved, N
        return out

    @staticmethod
    @_with_packs
    def backward(ctx, gout):
        shared, saved, N = ctx.shared, ctx.saved, ctx.N
        g = gout.contiguous()
        gparams = []
        with torch.cuda.device(g.device):
            for blk, (X, affine, Z, rows, K, cout) in reversed(list(zip(shared, saved))):
                tg, tb = _grad_target(blk.bn.weight), _grad_target(blk.bn.bias)
                into = (tg, tb) if (tg is not None and tb is not None) else None
                dZ, dgamma, dbeta = rows_bn_backward(g, Z, rows, cout, 1, N, 1, True, into=into)
                tw = _grad_target(blk.conv.weight)
                dw = rows_wgrad(dZ, X, cout, K, x_affine=affine, x_rows_per_stat=N,
                                into=None if tw is None else tw.view(cout, K), side=ctx.side)
                g = gemm_rows(dZ, blk.conv.weight.detach().reshape(cout, K), cout, K,
                              _packed("rows", blk.conv.weight))                           # gradient w.r.t. act(X)
                gparams = [None if dw is None else dw.reshape(blk.conv.weight.shape), dgamma, dbeta] + gparams
        return (g, None) + tuple(gparams)


def mlp_train(shared, x):
    return _MLPTrain.apply(x, shared, *mlp_params(shared))


# ---------------------------------------------------------------------------------------------
# the small heads (csrc/train_heads.hip): soft argmin, flow head, masked MAE -- one or two launches per direction where
# autograd makes 15-25 element-wise launches of each (a third of the step's dispatches were these)
# ---------------------------------------------------------------------------------------------
# 0: the ATen compositions (the A/B arm and what the tests compare the nodes with)
FUSED_HEADS = int(os.environ.get("PF_FUSED_HEADS", "1"))


class _SoftArgminTrain(torch.autograd.Function):
    """depth (B,1,H,W), prob map (B,1,H,W) of a filtered cost volume (B,D,H,W) (reference model.py:117-130): forward =
    row S's inference kernel, backward = pf_softargmin_backward_f32.  The probability map carries no gradient (the
    reference's loss never reads it, model.py:308-339)."""

    @staticmethod
    def forward(ctx, filtered, params):
        cost = filtered.detach().contiguous()
        depth, prob = pointflow.soft_argmin_params(cost, params)
        ctx.saved = (cost, params, depth)
        ctx.mark_non_differentiable(prob)
        ctx.set_materialize_grads(False)       # (else autograd fills a zero tensor for prob's gradient every step)
        return depth, prob

    @staticmethod
    def backward(ctx, gdepth, _gprob):
        if gdepth is None:
            return None, None
        cost, params, depth = ctx.saved
        B, D, H, W = cost.shape
        g = torch.empty_like(cost)
        with torch.cuda.device(cost.device):
            _lib.call("pf_softargmin_backward_f32", _lib.ptr(cost), _lib.ptr(params), _lib.ptr(depth),
                      _lib.ptr(gdepth.contiguous()), _lib.ptr(g), B, D, H * W, _lib.stream(),
                      algo_bytes=4.0 * B * H * W * (2 * D + 2))
        return g, None


def soft_argmin_train(filtered, params):
    return _SoftArgminTrain.apply(filtered, params)


class _FlowHeadTrain(torch.autograd.Function):
    """The 16 -> 1 convolution, the softmax over the five hypotheses and the expected offset (reference
    model.py:40-43, 218-227) on the MLP's (5*hw, 16) point-major output: (offset (hw,), prob (5, hw))."""

    @staticmethod
    def forward(ctx, act, weight, interval, hw):
        a = act.detach().contiguous()
        w = weight.detach().reshape(-1).contiguous()
        offset = torch.empty((hw,), dtype=_F32, device=a.device)
        prob = torch.empty((5, hw), dtype=_F32, device=a.device)
        with torch.cuda.device(a.device):
            _lib.call("pf_flow_head_train_f32", _lib.ptr(a), int(a.stride(0)), _lib.ptr(w), _lib.ptr(interval), hw,
                      _lib.ptr(offset), _lib.ptr(prob), _lib.stream(), algo_bytes=4.0 * hw * (5 * 16 + 6))
        ctx.saved = (a, w, interval, prob, weight)
        ctx.mark_non_differentiable(prob)
        ctx.set_materialize_grads(False)       # (no zero-filled gradient for prob)
        return offset, prob

    @staticmethod
    def backward(ctx, goffset, _gprob):
        if goffset is None:
            return None, None, None, None
        a, w, interval, prob, weight = ctx.saved
        hw = prob.shape[1]
        gact = torch.empty((5 * hw, 16), dtype=_F32, device=a.device)
        target = _grad_target(weight)
        gw = torch.empty((16,), dtype=_F32, device=a.device) if target is None else target.view(-1)
        nbytes = int(_lib.load().pf_flow_head_backward_workspace(hw))
        work = torch.empty((nbytes // 8,), dtype=torch.float64, device=a.device)
        with torch.cuda.device(a.device):
            _lib.call("pf_flow_head_backward_f32", _lib.ptr(a), int(a.stride(0)), _lib.ptr(w), _lib.ptr(interval),
                      _lib.ptr(prob), _lib.ptr(goffset.contiguous()), hw, _lib.ptr(gact), _lib.ptr(gw),
                      0 if target is None else 1, _lib.ptr(work), nbytes, _lib.stream(),
                      algo_bytes=4.0 * hw * (2 * 5 * 16 + 6))
        return gact, (gw.view(weight.shape) if target is None else None), None, None


def flow_head_supported(act, conv, D):
    return (FUSED_HEADS and act.is_cuda and act.dtype == _F32 and act.dim() == 2 and act.shape[1] == 16 and D == 5
            and act.shape[0] % 5 == 0 and conv.weight.numel() == 16)


def flow_head_train(act, conv, interval, hw):
    return _FlowHeadTrain.apply(act, conv.weight, interval, hw)


class _MaskedMAE(torch.autograd.Function):
    """weight * sum_b [ sum_{gt != 0} |pred - gt| / interval_b / (count_b + 1e-7) ] with the ground truth read at the
    nearest-resized positions (reference networks.py:170-181 under model.py:308-339's F.interpolate)."""

    @staticmethod
    def forward(ctx, pred, gt, interval, weight):
        p, t = pred.detach().contiguous(), gt.contiguous()
        B, _, h, w = p.shape
        H, W = t.shape[2:]
        loss = torch.empty((), dtype=_F32, device=p.device)
        coef = torch.empty((B,), dtype=_F32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.call("pf_masked_mae_f32", _lib.ptr(p), _lib.ptr(t), _lib.ptr(interval), B, h, w, H, W, float(weight),
                      _lib.ptr(loss), _lib.ptr(coef), _lib.stream(), algo_bytes=8.0 * B * h * w)
        ctx.saved = (p, t, coef)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        p, t, coef = ctx.saved
        B, _, h, w = p.shape
        H, W = t.shape[2:]
        g = torch.empty_like(p)
        with torch.cuda.device(p.device):
            _lib.call("pf_masked_mae_backward_f32", _lib.ptr(p), _lib.ptr(t), _lib.ptr(coef), _lib.ptr(gloss.contiguous()),
                      B, h, w, H, W, _lib.ptr(g), _lib.stream(), algo_bytes=12.0 * B * h * w)
        return g, None, None, None


def masked_mae_supported(pred, gt, interval):
    return (FUSED_HEADS and pred.is_cuda and pred.dtype == _F32 and gt.dtype == _F32 and interval.dtype == _F32
            and pred.dim() == 4 and pred.shape[1] == 1 and gt.dim() == 4 and gt.shape[1] == 1
            and interval.numel() == pred.shape[0] and interval.is_contiguous())


def masked_mae(pred, gt, interval, weight):
    return _MaskedMAE.apply(pred, gt, interval, weight)


# ---------------------------------------------------------------------------------------------
# warp + variance stages: coarse cost volume (reference model.py:79-111), flow feature assembly (model.py:153-204)
# ---------------------------------------------------------------------------------------------
def sort_pairs(keys, nkeys):
    """(order, start) of pf_sort_pairs_by_key: the pair ids grouped by key, ascending inside a group."""
    pairs = keys.numel()
    dev = keys.device
    order = torch.empty((max(pairs, 1),), dtype=torch.int32, device=dev)
    start = torch.empty((nkeys + 1,), dtype=torch.int32, device=dev)
    nbytes = int(_lib.load().pf_sort_pairs_workspace(max(pairs, 1), int(nkeys)))
    work = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
    _lib.call("pf_sort_pairs_by_key", _lib.ptr(keys), pairs, int(nkeys), _lib.ptr(order), _lib.ptr(start), _lib.ptr(work),
              nbytes, _lib.stream(), algo_bytes=16.0 * pairs + 8.0 * nkeys)
    return order, start


def _warp_backward(levels_cl, V, H, W, N, keys, fxy, dvar, ref_override, v0, planar=False):
    """gval + sorted gather: dmaps (V, H, W, ctot) channel-last (views < v0 left to the caller), and gval (V, N, ctot).
    dvar: (N, ctot) point-major rows, or (ctot, N) with ``planar``."""
    dev = dvar.device
    cs = [int(l.shape[3]) for l in levels_cl] + [0, 0]
    ctot = sum(cs)
    lv = list(levels_cl) + [None, None]
    order, start = sort_pairs(keys, V * (H + 1) * (W + 1))
    gval = torch.empty((V, N, ctot), dtype=_F32, device=dev)
    _lib.call("pf_variance_grad_f32", _lib.ptr(lv[0]), cs[0], _lib.ptr(lv[1]), cs[1], _lib.ptr(lv[2]), cs[2], V, H, W, N,
              _lib.ptr(keys), _lib.ptr(fxy), _lib.ptr(dvar), -1 if planar else int(dvar.stride(0)), int(bool(ref_override)),
              _lib.ptr(gval), _lib.stream(), algo_bytes=4.0 * (V * H * W * ctot + N * ctot * (1 + V)) + 12.0 * V * N)
    dmaps = torch.empty((V, H, W, ctot), dtype=_F32, device=dev)
    _lib.call("pf_warp_gather_f32", _lib.ptr(gval), _lib.ptr(fxy), _lib.ptr(order), _lib.ptr(start), V, int(v0), H, W, ctot,
              _lib.ptr(dmaps), _lib.stream(), algo_bytes=4.0 * ctot * (V * N + V * H * W) + 12.0 * V * N)
    return dmaps, gval


class _FlowFeaturesTrain(torch.autograd.Function):
    """Pyramid levels (V, c_l, h_l, w_l) + prior depth (h, w) -> the point-major feature rows (N, c1 + c2 + c3 + 24) and
    xyz (1, 3, N) of one PointFlow iteration (pf_flow_pyramid_f32 + pf_flow_features_f32, ratio 1); backward on
    csrc/warp_bwd.hip.  interval: one-element device tensor; cam: the packed camera block of this scale."""

    @staticmethod
    def forward(ctx, l1, l2, l3, depth, interval, cam, h, w):
        ctx.packs = _PACKS
        lv = [t.detach().contiguous() for t in (l1, l2, l3)]
        depth = depth.detach().contiguous()
        with torch.cuda.device(depth.device):
            levels = pointflow.flow_pyramid(lv, h, w)
            feature, xyz = pointflow.flow_features(levels, depth, interval, cam, h, w, 1)
        ctx.levels, ctx.depth, ctx.interval, ctx.cam, ctx.hw = levels, depth, interval, cam, (h, w)
        ctx.shapes = [tuple(t.shape) for t in lv]
        ctx.side = _on_side(2)
        ctx.mark_non_differentiable(xyz)
        ctx.set_materialize_grads(False)       # (no zero-filled gradient for xyz)
        return feature.view(feature.shape[1], feature.shape[2]), xyz

    @staticmethod
    @_with_packs
    def backward(ctx, dfeature, _dxyz):
        if dfeature is None:
            return (None,) * 8
        levels, depth, interval, cam = ctx.levels, ctx.depth, ctx.interval, ctx.cam
        h, w = ctx.hw
        V = levels[0].shape[0]
        N = 5 * h * w
        dev = depth.device
        g = dfeature.contiguous()
        cs = [int(l.shape[3]) for l in levels]
        ctot = sum(cs)
        with torch.cuda.device(dev):
            # the prior depth's gradient is what the chain (previous iteration, coarse stage) waits for ...
            ddepth = torch.empty((h, w), dtype=_F32, device=dev)
            _lib.call("pf_flow_depth_grad_f32", _lib.ptr(g), int(g.stride(0)), ctot, _lib.ptr(cam), h, w, _lib.ptr(ddepth),
                      _lib.stream(), algo_bytes=4.0 * N * 24 + 4.0 * h * w)
            # ... the pyramid levels' gradients only reach the flow tower's backward: with the tower on its own stream they
            # are produced there (autograd accumulates and consumes them on that stream: stream order makes it correct)
            side = ctx.side
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dev))
                for t in [g, depth] + list(levels):      # read there after this node (and its saved tensors) are gone
                    t.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                keys = torch.empty((V * N,), dtype=torch.int32, device=dev)
                fxy = torch.empty((V * N, 2), dtype=_F32, device=dev)
                _lib.call("pf_warp_taps_flow_f32", _lib.ptr(depth), _lib.ptr(interval), _lib.ptr(cam), V, h, w,
                          _lib.ptr(keys), _lib.ptr(fxy), _lib.stream(), algo_bytes=12.0 * V * N)
                dres, _ = _warp_backward(levels, V, h, w, N, keys, fxy, g, False, 0)
                outs, c0 = [], 0
                for c, shp in zip(cs, ctx.shapes):
                    dl = torch.empty(shp, dtype=_F32, device=dev)
                    _lib.call("pf_resize_bilinear_backward_f32", _lib.ptr(dres), ctot, c0, c, V, h, w, int(shp[2]),
                              int(shp[3]), _lib.ptr(dl), _lib.stream(), algo_bytes=4.0 * V * c * (h * w + shp[2] * shp[3]))
                    outs.append(dl)
                    c0 += c
        return outs[0], outs[1], outs[2], ddepth, None, None, None, None


def flow_features_supported(levels, depth, h, w):
    return (len(levels) == 3 and all(t.dim() == 4 and t.is_cuda and t.dtype == _F32 and t.shape[1] % 4 == 0 for t in levels)
            and depth.dim() == 2 and tuple(depth.shape) == (h, w) and levels[0].shape[0] <= 8)


def flow_features_train(levels, depth, interval, cam, h, w):
    return _FlowFeaturesTrain.apply(levels[0], levels[1], levels[2], depth, interval, cam, h, w)


class _CoarseVolumeTrain(torch.autograd.Function):
    """Coarse tower maps (V, C, FH, FW) of one scene -> cost volume (1, C, D*FH*FW) and the frustum points
    (pf_frustum_variance_cl_f32; view 0 contributes its un-warped map, reference model.py:103-106)."""

    @staticmethod
    def forward(ctx, maps, kinv, rinv, t, depths, K, E):
        ctx.packs = _PACKS
        from .utils.feature_fetcher import ChannelLast, frustum_variance, to_channel_last
        m = maps.detach().contiguous()
        with torch.cuda.device(m.device):
            cl = to_channel_last(m)                                         # (V, FH, FW, C)
            cost, world = frustum_variance(ChannelLast(cl.unsqueeze(0)), kinv, rinv, t, depths, K, E)
        ctx.cl = cl
        ctx.cams = tuple(x.detach().float().contiguous() for x in (kinv, rinv, t, depths, K, E))
        ctx.mark_non_differentiable(world)
        ctx.set_materialize_grads(False)       # (no zero-filled gradient for world)
        return cost, world

    @staticmethod
    @_with_packs
    def backward(ctx, dcost, _dworld):
        if dcost is None:
            return (None,) * 7
        cl = ctx.cl
        kinv, rinv, t, depths, K, E = ctx.cams
        V, FH, FW, C = cl.shape
        D = depths.shape[-1]
        N = D * FH * FW
        dev = cl.device
        with torch.cuda.device(dev):
            keys = torch.empty((V * N,), dtype=torch.int32, device=dev)
            fxy = torch.empty((V * N, 2), dtype=_F32, device=dev)
            _lib.call("pf_warp_taps_frustum_f32", _lib.ptr(kinv), _lib.ptr(rinv), _lib.ptr(t), _lib.ptr(depths), _lib.ptr(K),
                      _lib.ptr(E), V, FH, FW, D, 1, _lib.ptr(keys), _lib.ptr(fxy), _lib.stream(), algo_bytes=12.0 * V * N)
            dvar = dcost.reshape(C, N).contiguous()                           # channel-major as it is: no transposed copy
            dmaps, gval = _warp_backward([cl], V, FH, FW, N, keys, fxy, dvar, True, 1, planar=True)
            dmaps[0] = gval[0].view(D, FH * FW, C).sum(dim=0).view(FH, FW, C)   # the reference view: a sum over depth
            out = dmaps.permute(0, 3, 1, 2).contiguous()
        return out, None, None, None, None, None, None


def coarse_volume_train(maps, kinv, rinv, t, depths, K, E):
    return _CoarseVolumeTrain.apply(maps, kinv, rinv, t, depths, K, E)
