"""GPU parity tests of the training step's own backward kernels (Row Z; pointmvsnet_amd/train_ops.py, csrc/norm_bwd.hip,
csrc/conv_wgrad.hip, csrc/conv_dgrad.hip) against float64 autograd of the SAME ATen operators the reference
differentiates (reference nn/conv.py:24-35,62-77,108-121,197-210; networks.py:84-167; train.py:80), evaluated on the
same device.  Tolerances are float32-rounding class: a weight gradient is a sum over 10^4..10^6 positions of float32
products accumulated in float32 MFMA chains per block and added in a fixed order, so it is compared at 2e-5 of the
tensor's largest entry; data gradients and BatchNorm gradients at 1e-5 / 2e-5.  Every result is produced twice and
must be bit-identical (no atomics anywhere in the step).
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import report
from pointmvsnet_amd import _lib, networks, synthetic, train_ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _seeded(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("plane", [1, 0])
@pytest.mark.parametrize("N,C,S,sps,relu", [(3, 8, (40, 56), 1, True), (2, 16, (6, 10, 12), 2, True),
                                            (1, 64, (3, 4, 5), 1, False),
                                            # the one-launch form's five register depths (<= 1, 2, 4, 6, 8 pieces per
                                            # thread), three samples; and shapes it must leave to the two-launch form
                                            (3, 4, (6, 8, 10), 1, True), (3, 5, (64, 80), 1, True), (1, 3, (80, 128), 1, False),
                                            (1, 3, (128, 160), 1, True), (1, 2, (24, 32, 40), 1, True),
                                            (3, 3, (128, 160), 1, True),      # three big planes: the two-launch form

                                            (1, 2, (7, 9, 11), 1, True), (1, 2, (256, 160), 1, True)])
def test_bn_relu_backward_vs_float64_autograd(dev, N, C, S, sps, relu, plane, monkeypatch):
    """pf_bn_train_rows + pf_bn_bwd_reduce / _coeffs / _apply -- or, where a plane fits one block's registers,
    pf_bn_bwd_plane_f32 -- against autograd of F.batch_norm(training) (+ relu)."""
    monkeypatch.setattr(train_ops, "BN_BWD_PLANE", plane)
    y = _seeded((N, C) + S, dev, 1, 2.0) + 0.3
    g = _seeded((N, C) + S, dev, 2)
    bn = (torch.nn.BatchNorm2d if len(S) == 2 else torch.nn.BatchNorm3d)(C).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(1.0 + 0.2 * _seeded((C,), dev, 3))
        bn.bias.copy_(0.1 * _seeded((C,), dev, 4))
    spatial = y[0, 0].numel()
    T = 7
    # statistics partials as a producer's epilogue would leave them: T unequal slices of every (sample, channel)
    flat = y.reshape(N, C, spatial).double()
    cuts = [0] + sorted(set(int(spatial * (i + 1) / T) for i in range(T)))
    parts = torch.zeros((N, T, C, 2), dtype=torch.float64, device=dev)
    for t in range(len(cuts) - 1):
        seg = flat[:, :, cuts[t]:cuts[t + 1]]
        parts[:, t, :, 0] = seg.sum(-1)
        parts[:, t, :, 1] = (seg * seg).sum(-1)
    rows = train_ops.bn_train_rows(bn, parts, 0, C, float(sps * spatial), N, sps)
    z = train_ops.channel_affine(y, rows, sps, relu)
    dy, dgamma, dbeta = train_ops.bn_backward(g, y, rows, sps, relu)
    dy2, dgamma2, dbeta2 = train_ops.bn_backward(g, y, rows, sps, relu)
    assert torch.equal(dy, dy2) and torch.equal(dgamma, dgamma2) and torch.equal(dbeta, dbeta2)
    # reference: one F.batch_norm call per statistic group, float64
    yd = y.double().requires_grad_(True)
    w, b = bn.weight.double().detach().requires_grad_(True), bn.bias.double().detach().requires_grad_(True)
    outs = []
    for s in range(N // sps):
        o = F.batch_norm(yd[s * sps:(s + 1) * sps], None, None, w, b, True, 0.0, bn.eps)
        outs.append(torch.relu(o) if relu else o)
    zr = torch.cat(outs)
    zr.backward(g.double())
    e = dict(z=_rel(z, zr), dy=_rel(dy, yd.grad), dgamma=_rel(dgamma, w.grad), dbeta=_rel(dbeta, b.grad))
    report("bn_bwd_%dx%dx%d_sps%d" % (N, C, spatial, sps), **e)
    assert e["z"] < 2e-6 and e["dy"] < 1e-5 and e["dgamma"] < 2e-5 and e["dbeta"] < 2e-5, e


_WG_CASES = [
    # (N, Cout, Cin, in spatial, k, stride, affine)   -- ImageConv's eleven layer shapes (networks.py:89-110)
    (3, 8, 3, (40, 56), 3, 1, False), (3, 8, 8, (40, 56), 3, 1, True), (3, 16, 8, (40, 56), 5, 2, True),
    (2, 16, 16, (24, 40), 3, 1, True), (2, 32, 16, (24, 40), 5, 2, True), (2, 32, 32, (16, 24), 3, 1, True),
    (2, 64, 32, (16, 24), 5, 2, True), (3, 64, 64, (8, 12), 3, 1, False),
    # VolumeConv's (networks.py:133-147)
    (1, 8, 64, (8, 16, 24), 3, 1, False), (1, 16, 64, (8, 16, 24), 3, 2, False), (1, 32, 16, (8, 8, 12), 3, 2, True),
    (1, 64, 32, (4, 8, 12), 3, 2, False), (1, 64, 64, (2, 4, 6), 3, 1, False), (1, 16, 16, (4, 8, 12), 3, 1, False),
    (1, 32, 32, (4, 4, 6), 3, 1, False), (1, 1, 8, (8, 16, 24), 3, 1, False),
]


@pytest.mark.parametrize("N,Cout,Cin,sp,k,stride,affine", _WG_CASES)
def test_conv_weight_gradient_vs_float64_autograd(dev, N, Cout, Cin, sp, k, stride, affine):
    nd = len(sp)
    conv = F.conv2d if nd == 2 else F.conv3d
    x = _seeded((N, Cin) + sp, dev, 5)
    osp = tuple((s - 1) // stride + 1 for s in sp)
    dy = _seeded((N, Cout) + osp, dev, 6)
    sc = sh = None
    xin = x.double()
    if affine:
        sc = (1.0 + 0.3 * _seeded((N, Cin), dev, 7)).contiguous()
        sh = (0.2 * _seeded((N, Cin), dev, 8)).contiguous()
        view = (N, Cin) + (1,) * nd
        xin = torch.relu(xin * sc.double().view(view) + sh.double().view(view))
    w = torch.zeros((Cout, Cin) + (k,) * nd, dtype=torch.float64, device=dev, requires_grad=True)
    conv(xin, w, None, stride, k // 2).backward(dy.double())
    dw = train_ops.conv_wgrad(dy, x, (k,) * nd, stride, (k // 2,) * nd, None if sc is None else (sc, sh), 1)
    dw2 = train_ops.conv_wgrad(dy, x, (k,) * nd, stride, (k // 2,) * nd, None if sc is None else (sc, sh), 1)
    assert torch.equal(dw, dw2)
    err = _rel(dw, w.grad)
    report("conv_wgrad_%dd_%dto%d_k%ds%d" % (nd, Cin, Cout, k, stride), rel=err)
    assert dw.shape == w.shape and err < 2e-5, err


@pytest.mark.parametrize("late", [0, 2])
@pytest.mark.parametrize("N,Cout,Cin,sp", [(1, 8, 64, (8, 16, 24)), (1, 1, 8, (8, 16, 24)), (2, 8, 16, (24, 40)),
                                           (1, 8, 64, (5, 7, 19))])
def test_swapped_operand_weight_gradient_in_the_batched_reduce(dev, N, Cout, Cin, sp, late, monkeypatch):
    """Stride-1 layers with <= 8 output channels run pf_conv_wgrad_f32 with the operands swapped; inside the step the
    (Cin, Cout, reversed taps) partials are put into nn.ConvNd's order by pf_wgrad_reduce_batch_f32 (swapped), ADDED into the
    gradient slot, in one launch with a plain layer's partials.  Against float64 autograd, and against PF_WGRAD_SWAP=0.
    ``late``: the layers are queued until the node returns (0: _reduce_flush) or until the end of the backward (2, the
    default: flush_late) and issued together by pf_conv_wgrad_batch_f32 either way."""
    monkeypatch.setattr(train_ops, "WGRAD_LATE", late)
    nd = len(sp)
    conv = F.conv2d if nd == 2 else F.conv3d
    x = _seeded((N, Cin) + sp, dev, 11)
    dy = _seeded((N, Cout) + sp, dev, 12)
    x2 = _seeded((N, 16) + sp, dev, 13)                  # a second, plain layer (16 -> 16) in the same reduce launch
    dy2 = _seeded((N, 16) + sp, dev, 14)
    w = torch.zeros((Cout, Cin) + (3,) * nd, dtype=torch.float64, device=dev, requires_grad=True)
    w2 = torch.zeros((16, 16) + (3,) * nd, dtype=torch.float64, device=dev, requires_grad=True)
    conv(x.double(), w, None, 1, 1).backward(dy.double())
    conv(x2.double(), w2, None, 1, 1).backward(dy2.double())
    outs = {}
    for swap in (1, 0):
        monkeypatch.setattr(train_ops, "WGRAD_SWAP", swap)
        slot = torch.full(w.shape, 0.25, dtype=torch.float32, device=dev)      # "into": the bucket's slot, dw is ADDED
        slot2 = torch.full(w2.shape, -0.5, dtype=torch.float32, device=dev)
        with train_ops.direct_grads(True):
            assert train_ops.conv_wgrad(dy, x, (3,) * nd, 1, (1,) * nd, into=slot) is None
            assert train_ops.conv_wgrad(dy2, x2, (3,) * nd, 1, (1,) * nd, into=slot2) is None
            queue = train_ops._LATE["reduce"] if late else train_ops._REDUCE_PENDING
            assert len(queue) == 2 and bool(queue[0][6]) == bool(swap)
            if late:
                assert not train_ops._REDUCE_PENDING and len(train_ops._LATE["wgrad"]) == 2
                train_ops.flush_late()
            else:
                train_ops._reduce_flush()
            assert not train_ops._LATE["reduce"] and not train_ops._REDUCE_PENDING and not train_ops._WGRAD_DEFERRED
        outs[swap] = (slot - 0.25, slot2 + 0.5)
    e1, e0, e2 = _rel(outs[1][0], w.grad), _rel(outs[0][0], w.grad), _rel(outs[1][1], w2.grad)
    report("conv_wgrad_swapped_%dd_%dto%d" % (nd, Cin, Cout), rel=e1, rel_plain=e0, rel_neighbour=e2)
    assert e1 < 2e-5 and e0 < 2e-5 and e2 < 2e-5, (e1, e0, e2)
    assert torch.equal(outs[1][1], outs[0][1])                     # the plain neighbour does not notice


@pytest.mark.parametrize("Cin,Cout,sp", [(64, 32, (2, 4, 6)), (32, 16, (4, 8, 12)), (16, 8, (8, 8, 12))])
def test_transposed_conv_weight_gradient_vs_float64_autograd(dev, Cin, Cout, sp):
    """ConvTranspose3d(3, stride 2, pad 1, output_padding 1), VolumeConv's decoder (networks.py:141-143): the layer
    input sits on the coarse grid, dL/dy on the fine one; the result is in nn.ConvTranspose3d's (Cin, Cout, ...) order."""
    x = _seeded((1, Cin) + sp, dev, 9)
    dy = _seeded((1, Cout) + tuple(2 * s for s in sp), dev, 10)
    w = torch.zeros((Cin, Cout, 3, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
    F.conv_transpose3d(x.double(), w, None, 2, 1, 1).backward(dy.double())
    dw = train_ops.conv_wgrad(x, dy, (3, 3, 3), 2, (1, 1, 1))
    err = _rel(dw, w.grad)
    report("deconv_wgrad_%dto%d" % (Cin, Cout), rel=err)
    assert dw.shape == w.shape and err < 2e-5, err


@pytest.mark.parametrize("P,Cg,Cx,affine", [(5000, 64, 136, False), (4097, 64, 32, False), (3000, 128, 64, False),
                                            (6000, 64, 224, False), (2500, 64, 64, True), (2500, 16, 64, True)])
def test_rows_weight_gradient_vs_float64(dev, P, Cg, Cx, affine):
    """1x1 convolutions over point-major rows (EdgeConv's conv1 / conv2, the flow MLP) incl. strided row views."""
    gbuf = _seeded((P, Cg + 8), dev, 11)
    xbuf = _seeded((P, Cx + 12), dev, 12)
    g, x = gbuf[:, 4:4 + Cg], xbuf[:, 8:8 + Cx]
    xa = x.double()
    aff = None
    if affine:
        sc = (1.0 + 0.3 * _seeded((1, Cx), dev, 13)).contiguous()
        sh = (0.2 * _seeded((1, Cx), dev, 14)).contiguous()
        aff = (sc, sh)
        xa = torch.relu(xa * sc.double() + sh.double())
    ref = g.double().t() @ xa
    dw = train_ops.rows_wgrad(g, x, Cg, Cx, aff, P)
    assert torch.equal(dw, train_ops.rows_wgrad(g, x, Cg, Cx, aff, P))
    err = _rel(dw, ref)
    report("rows_wgrad_%dx%d" % (Cg, Cx), rel=err)
    assert err < 2e-5, err


@pytest.mark.parametrize("N,Cout,Cin,sp,k,stride", [(3, 8, 8, (40, 56), 3, 1), (2, 16, 8, (40, 56), 5, 2),
                                                    (2, 32, 16, (24, 40), 5, 2), (2, 64, 32, (16, 24), 5, 2),
                                                    (2, 64, 32, (10, 14), 5, 2), (2, 64, 64, (8, 12), 3, 1)])
def test_conv2d_data_gradient_vs_float64_autograd(dev, N, Cout, Cin, sp, k, stride):
    x = torch.zeros((N, Cin) + sp, dtype=torch.float64, device=dev, requires_grad=True)
    w = _seeded((Cout, Cin, k, k), dev, 15, 0.2)
    osp = tuple((s - 1) // stride + 1 for s in sp)
    dy = _seeded((N, Cout) + osp, dev, 16)
    F.conv2d(x, w.double(), None, stride, k // 2).backward(dy.double())
    dx = train_ops.conv2d_dgrad(dy, w, stride)
    err = _rel(dx, x.grad)
    report("conv2d_dgrad_%dto%d_k%ds%d" % (Cin, Cout, k, stride), rel=err)
    assert dx.shape == x.shape and err < 1e-5, err


def _grads(params):
    return [p.grad.detach().clone() for p in params]


def _oracle32_yardstick(fn, leaves64):
    """Worst per-tensor deviation (max |diff| / max |ref|) of a FLOAT32 evaluation of the reference's own composition (the
    oracle's functional restatement on plain ATen operators, same device) from the float64 gradients ``leaves64``:
    ``fn(dtype)`` evaluates it in ``dtype`` and returns the gradient list in the same order.  What a correct float32
    implementation deviates by at this size -- ReLU masks of pre-activations within rounding of zero and eleven BatchNorm
    layers amplify float32 rounding with the number of positions, so an absolute gate that holds on (64, 96) maps cannot
    hold on (512, 640) ones (measured on the emulator: 5.5e-3 on conv2.0's weight, the oracle itself 1.4e-2 on the
    whole step, profiles/r05_oracle_cfg4_yardstick.md)."""
    g32 = fn(torch.float32)
    return max(_rel(a, b) for a, b in zip(g32, leaves64))


AMBIGUOUS = 2e-5            # |pre-activation| below this: its ReLU mask is undetermined at float32 resolution


def _count_ambiguous_relu_inputs(ref):
    """Forward hooks on the BatchNorm layers of the float64 reference module: how many ReLU inputs lie within AMBIGUOUS of
    zero.  There a float32 and a float64 evaluation legitimately disagree about the mask, and ONE such flip moves the
    gradient entry at that position by its whole incoming gradient (measured on the emulator, VolumeConv at 48x64x80: one
    flip of 491 520 inputs of conv1_1 -- float64 pre-activation 8.8e-7 -- made dL/dy there -0.0088 instead of -0.52, 8e-2 of
    the largest entry, and everything upstream of it inherits a localized 5e-3: profiles/r05_emulator_runs.md).  With ~2e7
    ReLU inputs in a cfg-4 step a handful of them is always that close to zero; on (64, 96) maps usually none is."""
    box = {"n": 0}

    def hook(_m, _inp, out):
        box["n"] += int((out.detach().abs() < AMBIGUOUS).sum())

    handles = [m.register_forward_hook(hook) for m in ref.modules()
               if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d))]
    return box, handles


FLIP_GATE = 1e-1            # what flipped ReLU masks may move a gradient tensor by, relative to its largest entry


def test_image_tower_node_vs_float64_autograd(dev, size=(64, 96), yardstick=False):
    """The whole ImageConv tower (eleven layers, per-view BatchNorm statistics) as ONE autograd node against the ATen
    composition in float64: the three stage outputs, every parameter gradient, the running statistics.
    (tests/test_gpu_zz_train_cfg4.py calls this at (512, 640), the size of BASELINE configs[3], with ``yardstick``: the
    gradient gate is then tied to what the oracle's own float32 evaluation deviates by.)"""
    tower = networks.ImageConv(8)
    synthetic.seed_weights(tower, seed=3)
    tower = tower.to(dev).train()
    ref = networks.ImageConv(8)
    ref.load_state_dict({k: v.cpu() for k, v in tower.state_dict().items()})
    ref = ref.to(dev).double().train()
    img = _seeded((3, 3) + tuple(size), dev, 17)
    assert train_ops.tower_supported(tower, img)
    names = ("conv1", "conv2", "conv3")
    out = train_ops.tower_train(tower, img, names)
    gs = {n: _seeded(tuple(out[n].shape), dev, 18 + i) for i, n in enumerate(names)}
    sum((out[n] * gs[n]).sum() for n in names).backward()
    mine = _grads(tower.parameters())
    tower.zero_grad()
    out_b = train_ops.tower_train(tower, img, names)
    sum((out_b[n] * gs[n]).sum() for n in names).backward()
    for a, b in zip(mine, _grads(tower.parameters())):
        assert torch.equal(a, b)                                        # bit-reproducible
    amb, handles = _count_ambiguous_relu_inputs(ref)
    views = [ref(img[v:v + 1].double()) for v in range(3)]              # one call per view: per-view statistics
    for h_ in handles:
        h_.remove()
    sum((torch.cat([views[v][n] for v in range(3)]) * gs[n].double()).sum() for n in names).backward()
    worst = 0.0
    for n in names:
        e = _rel(out[n], torch.cat([views[v][n] for v in range(3)]))
        worst = max(worst, e)
        assert e < 2e-5, (n, e)
    errs = sorted(((_rel(a, p.grad), k) for a, (k, p) in zip(mine, ref.named_parameters())), reverse=True)
    gate, yard = 2e-4, 0.0
    if yardstick:
        from oracle import pointflow_oracle as O
        sd = {"t." + k: v.detach() for k, v in tower.state_dict().items()}
        pnames = [k for k, _ in tower.named_parameters()]

        def oracle_grads(dtype):
            leaves = {k: (v.to(dtype).clone().requires_grad_(True) if k[2:] in pnames else v.clone()) for k, v in sd.items()}
            outs = [O.image_conv(img[v:v + 1].to(dtype), leaves, "t") for v in range(3)]
            sum((torch.cat([outs[v][n] for v in range(3)]) * gs[n].to(dtype)).sum() for n in names).backward()
            return [leaves["t." + k].grad for k in pnames]

        yard = _oracle32_yardstick(oracle_grads, [p.grad for p in ref.parameters()])
        gate = max(gate, 3.0 * yard, FLIP_GATE if amb["n"] else 0.0)
    report("tower_node" if tuple(size) == (64, 96) else "tower_node_%dx%d" % tuple(size), out_rel=worst, worst_grad_rel=errs[0][0],
           median_grad_rel=errs[len(errs) // 2][0], oracle32_worst_grad_rel=yard, ambiguous_relu_inputs=amb["n"])
    assert errs[0][0] < gate, (errs[:5], yard, amb)
    for (k, a), (_, b) in zip(tower.named_buffers(), ref.named_buffers()):
        if "num_batches" in k:
            assert int(a) == 2 * int(b) == 6, k                          # two forwards of three views here
    # only "conv3" wanted (the coarse tower): same gradients for the layers that feed it alone
    tower.zero_grad()
    o3 = train_ops.tower_train(tower, img, ("conv3",))["conv3"]
    (o3 * gs["conv3"]).sum().backward()
    assert all(p.grad is not None for p in tower.parameters())


def test_volume_conv_node_vs_float64_autograd(dev, size=(16, 32, 40), yardstick=False):
    """VolumeConv as ONE autograd node against the float64 ATen module (tests/test_gpu_zz_train_cfg4.py calls this at
    (48, 64, 80), the size of BASELINE configs[3], with ``yardstick``: the gradient gates are then tied to what the
    oracle's own float32 evaluation deviates by, see _oracle32_yardstick)."""
    vc = networks.VolumeConv(64, 8)
    synthetic.seed_weights(vc, seed=4)
    vc = vc.to(dev).train()
    ref = networks.VolumeConv(64, 8)
    ref.load_state_dict({k: v.cpu() for k, v in vc.state_dict().items()})
    ref = ref.to(dev).double().train()
    cost = (_seeded((1, 64) + tuple(size), dev, 21).abs()).requires_grad_(True)
    assert train_ops.volume_supported(vc, cost)
    out = train_ops.volume_train(vc, cost)
    g = _seeded(tuple(out.shape), dev, 22)
    (out * g).sum().backward()
    mine, gx = _grads(vc.parameters()), cost.grad.detach().clone()
    vc.zero_grad()
    cost.grad = None
    (train_ops.volume_train(vc, cost) * g).sum().backward()
    assert torch.equal(gx, cost.grad)
    for a, b in zip(mine, _grads(vc.parameters())):
        assert torch.equal(a, b)
    cd = cost.detach().double().requires_grad_(True)
    amb, handles = _count_ambiguous_relu_inputs(ref)
    rout = ref(cd)
    for h_ in handles:
        h_.remove()
    (rout * g.double()).sum().backward()
    errs = sorted(((_rel(a, p.grad), k) for a, (k, p) in zip(mine, ref.named_parameters())), reverse=True)
    e_out, e_x = _rel(out, rout), _rel(gx, cd.grad)
    # how much of d(cost) deviates at all: a flipped mask is a LOCAL event (its receptive field), a wrong kernel is not
    frac_x = float(((gx.double() - cd.grad).abs() > 1e-4 * cd.grad.abs().max()).float().mean())
    gate_x, gate_w, yard = 1e-4, 2e-4, (0.0, 0.0)
    if yardstick:
        from oracle import pointflow_oracle as O
        sd = {"v." + k: v.detach() for k, v in vc.state_dict().items()}
        pnames = [k for k, _ in vc.named_parameters()]

        def oracle_grads(dtype):
            leaves = {k: (v.to(dtype).clone().requires_grad_(True) if k[2:] in pnames else v.clone()) for k, v in sd.items()}
            c = cost.detach().to(dtype).requires_grad_(True)
            (O.volume_conv(c, leaves, "v") * g.to(dtype)).sum().backward()
            return [c.grad] + [leaves["v." + k].grad for k in pnames]

        g32 = oracle_grads(torch.float32)
        yard = (_rel(g32[0], cd.grad), max(_rel(a, p.grad) for a, p in zip(g32[1:], ref.parameters())))
        flip = FLIP_GATE if amb["n"] else 0.0
        gate_x, gate_w = max(gate_x, 3.0 * yard[0], flip), max(gate_w, 3.0 * yard[1], flip)
    report("volume_node" if tuple(size) == (16, 32, 40) else "volume_node_%dx%dx%d" % tuple(size), out_rel=e_out, dcost_rel=e_x,
           worst_grad_rel=errs[0][0], median_grad_rel=errs[len(errs) // 2][0], oracle32_dcost_rel=yard[0],
           oracle32_worst_grad_rel=yard[1], ambiguous_relu_inputs=amb["n"], dcost_frac_beyond_1e4=frac_x)
    assert e_out < 2e-5 and e_x < gate_x and frac_x < 1e-2, (e_out, e_x, frac_x, yard, amb)
    assert errs[0][0] < gate_w and errs[len(errs) // 2][0] < 2e-5, (errs[:5], errs[len(errs) // 2], yard, amb)


def test_edge_chain_xcd_band_order_is_a_pure_relabelling(dev):
    """Round 6: with the lattice's plane shape as a hint the EdgeConv gather passes -- forward statistics / apply, backward
    reduce and inverted-list gather -- number their blocks so that an XCD owns a band of pixel rows in every plane
    (csrc/edgeconv.hip: xcd_tile; here 5 planes of 16 x 32 points: 8 tiles per plane, band = 1).  A relabelling of the grid
    only: outputs, the input gradient and every parameter gradient are the SAME BITS as without the hint, and the
    BatchNorm running statistics too (the partial rows stay in tile order)."""
    from pointmvsnet_amd.model import PointMVSNet
    from pointmvsnet_amd.utils.torch_utils import get_knn_3d
    D, h, w = 5, 16, 32
    N = D * h * w
    xyz = _seeded((1, 3, D, h, w), dev, 31)
    idx = get_knn_3d(xyz, 5, knn=16)
    g = None
    got = []
    for plane_hw in (None, (h, w)):
        net = PointMVSNet()
        synthetic.seed_weights(net, seed=0)
        net = net.to(dev).train()
        feat = _seeded((N, 136), dev, 32).requires_grad_(True)
        edges = train_ops.edge_chain_train(net.flow_edge_conv, feat, idx, plane_hw=plane_hw)
        g = _seeded(tuple(edges.shape), dev, 33) if g is None else g
        (edges * g).sum().backward()
        params = list(net.flow_edge_conv.parameters())
        got.append([edges.detach().clone(), feat.grad.detach().clone()] + _grads(params)
                   + [b.detach().clone() for b in net.flow_edge_conv.buffers()])
    assert len(got[0]) == len(got[1])
    for a, b in zip(*got):
        assert torch.equal(a, b)
    assert _lib.status() == 0


def test_edge_chain_and_mlp_nodes_vs_composed_operators(dev):
    """EdgeConv x3 + SharedMLP on point-major rows (two nodes) against a float64 composition on the same device: the
    reference side builds its modules in double precision (their forwards are then plain ATen: F.conv1d, an explicit
    torch.gather, F.batch_norm; nothing of this package's kernels runs there -- the neighbour indices are an input of
    both sides): outputs, input gradient, parameter gradients.  The same comparison with the ATen calls written out
    function by function is tests/test_gpu_zz_train_cfg4.py::test_edge_chain_and_mlp_nodes_vs_float64_functional."""
    from pointmvsnet_amd.model import PointMVSNet
    from pointmvsnet_amd.utils.torch_utils import get_knn_3d
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    D, h, w = 5, 16, 24
    N = D * h * w
    xyz = _seeded((1, 3, D, h, w), dev, 23)
    idx = get_knn_3d(xyz, 5, knn=16)
    feat = _seeded((N, 136), dev, 24).requires_grad_(True)
    assert train_ops.edge_chain_supported(net.flow_edge_conv, feat, idx)
    edges = train_ops.edge_chain_train(net.flow_edge_conv, feat, idx)
    assert train_ops.mlp_supported(net.flow_mlp[0], edges)
    act = train_ops.mlp_train(net.flow_mlp[0], edges)
    g = _seeded(tuple(act.shape), dev, 25)
    (act * g).sum().backward()
    params = list(net.flow_edge_conv.parameters()) + list(net.flow_mlp[0].parameters())
    mine, gfeat = _grads(params), feat.grad.detach().clone()
    for p in params:
        p.grad = None
    feat.grad = None
    act_b = train_ops.mlp_train(net.flow_mlp[0], train_ops.edge_chain_train(net.flow_edge_conv, feat, idx))
    (act_b * g).sum().backward()
    assert torch.equal(act, act_b) and torch.equal(gfeat, feat.grad)
    for a, b in zip(mine, _grads(params)):
        assert torch.equal(a, b)
    # composed: float64 modules, explicit gather
    ref = PointMVSNet()
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    ref = ref.to(dev).double().train()
    fd = feat.detach().double().requires_grad_(True)
    x = fd.t().unsqueeze(0)
    outs = []
    for m in ref.flow_edge_conv:
        k = idx.shape[2]
        l, e = m.conv1(x), m.conv2(x)
        nb = torch.gather(e.unsqueeze(3).expand(-1, -1, -1, k), 2, idx.unsqueeze(1).expand(-1, e.shape[1], -1, -1))
        central = l.unsqueeze(-1).expand(-1, -1, -1, k)
        edge = torch.cat([central, nb - central], dim=1) if m.concat else nb - central
        x = torch.relu(m.bn(edge)).mean(dim=3)
        outs.append(x)
    y = torch.cat(outs, dim=1)
    for blk in ref.flow_mlp[0]:
        y = torch.relu(blk.bn(blk.conv(y)))
    ract = y[0].t()
    (ract * g.double()).sum().backward()
    rparams = list(ref.flow_edge_conv.parameters()) + list(ref.flow_mlp[0].parameters())
    errs = sorted(((_rel(a, p.grad), i) for i, (a, p) in enumerate(zip(mine, rparams))), reverse=True)
    e_act, e_x = _rel(act, ract), _rel(gfeat, fd.grad)
    report("edge_chain_mlp_nodes", act_rel=e_act, dfeature_rel=e_x, worst_grad_rel=errs[0][0],
           median_grad_rel=errs[len(errs) // 2][0])
    # the forward agrees to float32 rounding; in the backward a float32 and a float64 evaluation legitimately disagree
    # about the ReLU mask of the pre-activations within rounding of zero (a few per thousand of the 16 N edge values,
    # tests/test_gpu_backward_cfg4.py), which moves single gradient entries: measured 6e-4 / 4e-3 / median 3e-4
    assert e_act < 2e-5 and e_x < 3e-3 and errs[0][0] < 1e-2 and errs[len(errs) // 2][0] < 1e-3, (e_act, e_x, errs[:5])


def _tiny_plan(dev):
    from pointmvsnet_amd.model import PointMVSNet
    data, img_scales, inter_scales = synthetic.make_config("tiny", train_intrinsics=True)
    batch = {k: v.to(dev) for k, v in data.items()}
    batch["cam_params_list_host"] = data["cam_params_list"]
    batch["mean_host"], batch["std_host"] = data["mean"], data["std"]
    net = PointMVSNet().to(dev).train()
    tplan = net.make_train_plan(batch, img_scales, inter_scales, isTest=False)
    torch.cuda.synchronize()
    return net, tplan, data["img_list"].shape[1]


@pytest.mark.parametrize("it,h,w", [(0, 16, 24), (1, 32, 48)])
def test_flow_feature_node_vs_composed_operators(dev, it, h, w):
    """Feature assembly of a PointFlow iteration (resize + warp + variance + xyz; reference model.py:153-204) as one
    node -- backward on csrc/warp_bwd.hip, no atomics -- against the reference's composition on differentiable
    operators (F.interpolate, the HIP FeatureFetcher with its atomic scatter, ATen): the feature rows, the gradients
    w.r.t. the three pyramid levels and w.r.t. the prior depth; and bit-reproducibility."""
    net, tplan, V = _tiny_plan(dev)
    H, W = 128, 192
    pyr = {n: _seeded((1, V, c, H // s, W // s), dev, 30 + i).requires_grad_(True)
           for i, (n, c, s) in enumerate((("conv1", 16, 2), ("conv2", 32, 4), ("conv3", 64, 8)))}
    depth = (600.0 + 40.0 * _seeded((1, 1, h, w), dev, 34)).requires_grad_(True)
    gfeat = _seeded((5 * h * w, 136), dev, 35)
    levels = [pyr[n][0] for n in ("conv1", "conv2", "conv3")]
    assert train_ops.flow_features_supported(levels, depth[0, 0], h, w)
    pack = tplan.d("pack%d" % it)[0]
    runs = []
    for _ in range(2):
        rows, xyz = train_ops.flow_features_train(levels, depth[0, 0], pack[-1:], pack, h, w)
        (rows * gfeat).sum().backward()
        runs.append([rows.detach().clone(), xyz.detach().clone(), depth.grad.clone()] + [pyr[n].grad.clone() for n in pyr])
        depth.grad = None
        for n in pyr:
            pyr[n].grad = None
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    feature, xyz_ref = net._assemble_autograd(pyr, depth, tplan, it, h, w)          # (1,136,5,hw), (1,3,5,h,w)
    rows_ref = feature.view(136, 5 * h * w).t()
    (rows_ref * gfeat).sum().backward()
    scale = float(rows_ref.abs().max())
    e = dict(rows=float((runs[0][0] - rows_ref).abs().max()) / scale, xyz=_rel(runs[0][1].view(-1), xyz_ref.reshape(-1)),
             ddepth=_rel(runs[0][2], depth.grad))
    for i, n in enumerate(pyr):
        e["d" + n] = _rel(runs[0][3 + i], pyr[n].grad)
    report("flow_feature_node_it%d" % it, **e)
    assert e["rows"] < 2e-5 and e["xyz"] < 1e-5, e
    assert e["ddepth"] < 1e-4 and max(e["dconv1"], e["dconv2"], e["dconv3"]) < 1e-4, e


def test_coarse_volume_node_vs_composed_operators(dev):
    """Coarse cost volume (reference model.py:79-111) as one node against the composition (frustum by matmul, the HIP
    FeatureFetcher with its atomic scatter, ATen variance): the volume, the gradient w.r.t. the tower maps."""
    net, tplan, V = _tiny_plan(dev)
    C, FH, FW, D = 64, 16, 24, tplan.D
    maps = _seeded((V, C, FH, FW), dev, 40).requires_grad_(True)
    gcost = _seeded((1, C, D * FH * FW), dev, 41)
    args = (tplan.d("Kinv0"), tplan.d("Rinv0"), tplan.d("t0"), tplan.d("depths"), tplan.d("K_coarse"), tplan.d("ext"))
    runs = []
    for _ in range(2):
        cost, world = train_ops.coarse_volume_train(maps, *args)
        (cost * gcost).sum().backward()
        runs.append((cost.detach().clone(), world.detach().clone(), maps.grad.clone()))
        maps.grad = None
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    grid = net._pixel_grid(FH, FW, dev).view(1, 1, 3, -1)
    uv = torch.matmul(tplan.d("Kinv0"), grid)
    cam_points = (uv.unsqueeze(3) * tplan.d("depths").view(1, 1, 1, D, 1)).view(1, 1, 3, -1)
    world_ref = torch.matmul(tplan.d("Rinv0"), cam_points - tplan.d("t0")).transpose(1, 2).contiguous().view(1, 3, -1)
    cost_ref = net._coarse_cost_autograd(maps.unsqueeze(0), world_ref, tplan.d("K_coarse"), tplan.d("ext"), D)
    (cost_ref * gcost).sum().backward()
    e = dict(cost=float((runs[0][0] - cost_ref).abs().max()) / float(cost_ref.abs().max()),
             world=_rel(runs[0][1], world_ref), dmaps=_rel(runs[0][2], maps.grad))
    report("coarse_volume_node", **e)
    assert e["cost"] < 2e-5 and e["world"] < 1e-6 and e["dmaps"] < 1e-4, e


def test_step_weight_packs_equal_the_torch_built_layouts(dev):
    """train_packs.TrainPacks: every layout the step's kernels read their weights in, produced by ONE affine-gather launch,
    is bit-equal to the same layout built with torch operators (the packers of pointflow.py / train_ops.py)."""
    from pointmvsnet_amd import pointflow
    from pointmvsnet_amd.model import PointMVSNet
    from pointmvsnet_amd.train_packs import TrainPacks
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=5)
    net = net.to(dev).train()
    packs = TrainPacks(net)
    packs.run()
    torch.cuda.synchronize()
    assert not packs.stale()
    flip_t = train_ops._flip_t
    checked = 0

    def c3(w):
        cout, cin = w.shape[:2]
        ncp = (cout + 15) // 16 * 16
        wp = torch.zeros((cin // 4, 27, 4, ncp), device=dev)
        wp[..., :cout] = w.detach().permute(1, 2, 3, 4, 0).reshape(cin // 4, 4, 27, cout).transpose(1, 2)
        return wp

    def c3b(w):
        cin = w.shape[1]
        return w.detach().permute(2, 3, 4, 1, 0).reshape(3, 3, 3, cin // 16, 4, 4, 64).permute(0, 1, 2, 3, 4, 6, 5).contiguous()

    def d3b(w):
        cin, cout = w.shape[:2]
        return w.detach().permute(2, 3, 4, 0, 1).reshape(27, cin // 16, 4, 4, cout).permute(0, 1, 2, 4, 3).contiguous()

    for tower in (net.coarse_img_conv, net.flow_img_conv):
        for i, (_, _, conv, _) in enumerate(train_ops._tower_blocks(tower)):
            W = conv.weight
            assert torch.equal(packs.get("c2w", W), pointflow._pack_conv2d_wide(W).view_as(packs.get("c2w", W)))
            checked += 1
            if i > 0 and conv.stride[0] == 1:
                ref = pointflow._pack_conv2d_wide(flip_t(W))
                assert torch.equal(packs.get("c2w_dg", W), ref.view_as(packs.get("c2w_dg", W)))
                checked += 1
            elif i > 0:
                co, ci, k, _ = W.shape
                ref = torch.zeros((co // 4, k * k, 4, (ci + 15) // 16 * 16), device=dev)
                ref[..., :ci] = W.detach().reshape(co // 4, 4, ci, k * k).permute(0, 3, 1, 2)
                assert torch.equal(packs.get("d2_dg", W), ref)
                checked += 1
    vc = net.coarse_vol_conv
    with pointflow.no_pack_cache():
        assert torch.equal(packs.get("c3p", vc.conv0_1.conv.weight), pointflow.pack_conv3d_weight_pair(vc.conv0_1.conv.weight))
    for name in ("conv1_0", "conv2_0", "conv1_1", "conv2_1", "conv6_0", "conv5_0"):
        W = getattr(vc, name).conv.weight
        assert torch.equal(packs.get("c3", W), c3(W)), name
    for name in ("conv1_1", "conv2_1"):
        W = getattr(vc, name).conv.weight
        assert torch.equal(packs.get("c3_dg", W), c3(flip_t(W))), name
    W01 = vc.conv0_1.conv.weight
    for h in (0, 1):
        assert torch.equal(packs.get("c3_dg%d" % h, W01), c3(flip_t(W01)[32 * h:32 * h + 32].contiguous()))
    for name in ("conv3_0", "conv3_1", "conv4_0"):
        W = getattr(vc, name).conv.weight
        assert torch.equal(packs.get("c3b", W), c3b(W)), name
    assert torch.equal(packs.get("c3b_dg", vc.conv3_1.conv.weight), c3b(flip_t(vc.conv3_1.conv.weight)))
    for name in ("conv4_0", "conv3_0"):
        W = getattr(vc, name).conv.weight
        assert torch.equal(packs.get("d3b", W), d3b(W)), name
    W62 = vc.conv6_2.weight
    assert torch.equal(packs.get("c1_dg", W62), W62.detach().flip(2, 3, 4).reshape(W62.shape[1], 27))
    for e in net.flow_edge_conv:
        wt, cout = packs._dst[("wt", id(e.conv1.weight), id(e.conv2.weight))]
        ref, rc = pointflow._pack_weight_t(e.conv1.weight, e.conv2.weight)
        assert cout == rc and torch.equal(wt, ref)
        C, K = e.conv1.weight.shape[:2]
        wcat = torch.cat([e.conv1.weight.detach().reshape(C, K), e.conv2.weight.detach().reshape(C, K)], dim=0)
        for col, width, chunk in packs.get("rows", e.conv1.weight):
            assert torch.equal(chunk[:, :width], wcat[:, col:col + width]) and float(chunk[:, width:].abs().sum()) == 0.0
    for blk in net.flow_mlp[0]:
        W = blk.conv.weight
        wt, cout = packs._dst[("wt", id(W))]
        ref, rc = pointflow._pack_weight_t(W)
        assert cout == rc and torch.equal(wt, ref)
        for col, width, chunk in packs.get("rows", W):
            assert torch.equal(chunk[:, :width], W.detach().reshape(W.shape[0], -1)[:, col:col + width])
    # an optimizer-style in-place update is picked up by the next run(); moved storage is reported
    with torch.no_grad():
        W01.mul_(2.0)
    packs.run()
    assert torch.equal(packs.get("c3_dg0", W01), c3(flip_t(W01)[:32].contiguous()))
    W01.data = W01.data.clone()
    assert packs.stale()
    assert checked >= 40


def test_soft_argmin_node_vs_float64_autograd(dev):
    """train_ops.soft_argmin_train (row S's kernel + pf_softargmin_backward_f32) against the reference's composition
    softmax(-cost) -> sum_k linspace_k p_k (model.py:117-124) in float64; the probability map against
    functions.get_propability_map through the inference kernel's own tests (tests/test_gpu_ops.py)."""
    B, D, H, W = 2, 48, 16, 20
    cost = _seeded((B, D, H, W), dev, 31, 2.0)
    start = torch.tensor([425.0, 500.0], device=dev)
    interval = torch.tensor([2.5, 3.0], device=dev)
    end = start + (D - 1) * interval
    params = torch.stack([start, end, interval], dim=1).contiguous()
    g = _seeded((B, 1, H, W), dev, 32)
    x = cost.clone().requires_grad_(True)
    depth, prob = train_ops.soft_argmin_train(x, params)
    (depth * g).sum().backward()
    x2 = cost.clone().requires_grad_(True)
    depth2, _ = train_ops.soft_argmin_train(x2, params)
    (depth2 * g).sum().backward()
    assert torch.equal(depth, depth2) and torch.equal(x.grad, x2.grad)
    xr = cost.double().requires_grad_(True)
    z = torch.stack([torch.linspace(float(start[b]), float(end[b]), D, dtype=torch.float64, device=dev) for b in range(B)])
    p = F.softmax(-xr, dim=1)
    dr = (z.view(B, D, 1, 1) * p).sum(dim=1, keepdim=True)
    (dr * g.double()).sum().backward()
    e_d, e_g = _rel(depth, dr), _rel(x.grad, xr.grad)
    report("train_soft_argmin_node", depth=e_d, grad=e_g)
    assert not prob.requires_grad and prob.shape == depth.shape
    assert e_d < 2e-6 and e_g < 1e-5, (e_d, e_g)


def test_flow_head_node_vs_float64_autograd(dev):
    """train_ops.flow_head_train against (act * w).sum -> softmax(-flow) over the five hypotheses -> sum prob * length
    (reference model.py:40-43, 218-227) in float64: offset, probabilities, gradient rows and the 16 weight gradients."""
    hw = 37 * 41
    conv = torch.nn.Conv1d(16, 1, 1, bias=False).to(dev)
    act = torch.relu(_seeded((5 * hw, 16), dev, 41)) + 0.01
    interval = torch.tensor([1.9], device=dev)
    g = _seeded((hw,), dev, 42)
    a = act.clone().requires_grad_(True)
    offset, prob = train_ops.flow_head_train(a, conv, interval, hw)
    conv.weight.grad = None
    (offset * g).sum().backward()
    gw1, ga1 = conv.weight.grad.clone(), a.grad.clone()
    a2 = act.clone().requires_grad_(True)
    conv.weight.grad = None
    offset2, _ = train_ops.flow_head_train(a2, conv, interval, hw)
    (offset2 * g).sum().backward()
    assert torch.equal(offset, offset2) and torch.equal(ga1, a2.grad) and torch.equal(gw1, conv.weight.grad)
    ar = act.double().requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    flow = (ar * wr.view(1, -1)).sum(dim=1).view(5, hw)
    pr = F.softmax(-flow, dim=0)
    length = torch.tensor([-2.0, -1.0, 0.0, 1.0, 2.0], dtype=torch.float64, device=dev).view(5, 1) * interval.double()
    offr = (pr * length).sum(dim=0)
    (offr * g.double()).sum().backward()
    errs = dict(offset=_rel(offset, offr), prob=_rel(prob, pr), gact=_rel(ga1, ar.grad), gw=_rel(gw1, wr.grad))
    report("train_flow_head_node", **errs)
    assert errs["offset"] < 5e-6 and errs["prob"] < 2e-6 and errs["gact"] < 1e-5 and errs["gw"] < 1e-5, errs


@pytest.mark.parametrize("B,h,w,H,W", [(1, 128, 160, 512, 640), (2, 7, 9, 30, 37), (1, 64, 80, 512, 640)])
def test_masked_mae_node_vs_float64_composition(dev, B, h, w, H, W):
    """train_ops.masked_mae against F.interpolate(gt) (nearest) + MAELoss (reference networks.py:170-181,
    model.py:308-339) in float64, ground truth with holes (zeros), a non-integer resize ratio among the cases."""
    gt = (_seeded((B, 1, H, W), dev, 51).abs() * 100 + 400)
    gt = gt * (_seeded((B, 1, H, W), dev, 52) > -0.5).float()                   # ~30 % invalid
    pred = F.interpolate(gt, (h, w)) + _seeded((B, 1, h, w), dev, 53, 3.0)
    pred[0, 0, 0, 0] = F.interpolate(gt, (h, w))[0, 0, 0, 0]                     # an exact hit: sign(0) = 0
    interval = (torch.rand(B, device=dev) + 2.0).contiguous()
    weight = 1.0 / (3 * 0.75)
    p = pred.clone().requires_grad_(True)
    loss = train_ops.masked_mae(p, gt, interval, weight)
    (loss * 1.7).backward()
    p2 = pred.clone().requires_grad_(True)
    loss2 = train_ops.masked_mae(p2, gt, interval, weight)
    (loss2 * 1.7).backward()
    assert torch.equal(loss, loss2) and torch.equal(p.grad, p2.grad)
    pr = pred.double().requires_grad_(True)
    target = F.interpolate(gt, (h, w)).double()
    ref = networks.MAELoss()(pr, target, interval.double()) * weight
    (ref * 1.7).backward()
    e_l, e_g = abs(float(loss) - float(ref)) / abs(float(ref)), _rel(p.grad, pr.grad)
    report("train_masked_mae_node_%dx%d" % (h, w), loss=e_l, grad=e_g)
    assert loss.shape == () and e_l < 2e-7 and e_g < 2e-7, (e_l, e_g)
