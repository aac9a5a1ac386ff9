"""Row Z at ITS OWN size: every backward launch of BASELINE configs[3]'s training step (640x512, 3 views, 48 planes;
flow-1 on 25 600 points, flow-2 on 102 400) at the EXACT shape the step launches it with, against float64 autograd of
the ATen operator the reference differentiates (reference train.py:72-82, networks.py:84-167,
nn/conv.py:24-35,62-77,108-121,197-210), on the same device.

Why this file exists: the weight-gradient kernel's launch plan is a function of the shape (tile candidates, row tiles
per block, channel blocks, position splits, the big-tile / 128-point modes that only switch on above a size;
csrc/conv_wgrad.hip make_plan), and tests/test_gpu_train_ops.py runs at (40, 56) images and (8, 16, 24) volumes.
Here

  * ``test_step_launches_exactly_the_tabulated_shapes`` runs one real cfg-4 step with the C ABI and the data-gradient /
    BatchNorm-backward helpers recorded and asserts that the SET of launched shapes equals the tables below -- so the
    per-shape tests cover the step, all of it and nothing else;
  * every table row is then checked alone against float64 (same gates as the small-shape tests), twice, bit for bit,
    and its launch plan (pf_conv_wgrad_plan / pf_rows_wgrad_plan: a pure function of the shape) goes to the parity
    report: the plan tested is the plan the step uses because it is the same shape on the same device;
  * the tower and VolumeConv NODES run at (3, 3, 512, 640) / (1, 64, 48, 64, 80) against the float64 ATen modules.

Further down: the whole step at this size against the oracle (float32 and float64 on the host; the function of
tests/test_gpu_model.py with cfg = "cfg4"), its bit-reproducibility and graphed == eager; and the float64 arms of the
node tests whose round-4 reference side ran on this package's own operators (flow features, coarse volume; the EdgeConv
chain with every ATen call written out).

The file sorts last on purpose: these are the longest tests of the suite.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import report
from pointmvsnet_amd import _lib, networks, pointflow, synthetic, train_ops

# Round 5 wrote this file without GPU access and marked it xfail(strict=False); the driver's round-end run was its first
# hardware execution and every test XPASSed (GPUTEST_r05: 340 passed, 82 xpassed).  Round 6 removed the marker: these
# tests gate like every other one.
pytestmark = [pytest.mark.gpu]

V, H, W, D = 3, 512, 640, 48
P1, P2 = 5 * 64 * 80, 5 * 128 * 160          # points of PointFlow iteration 1 / 2

# ImageConv (networks.py:89-110): (Cout, Cin, input (H, W), k, stride, the input carries a pending BatchNorm + ReLU)
TOWER = [
    (8, 3, (512, 640), 3, 1, False), (8, 8, (512, 640), 3, 1, True),
    (16, 8, (512, 640), 5, 2, True), (16, 16, (256, 320), 3, 1, True),
    (32, 16, (256, 320), 5, 2, True), (32, 32, (128, 160), 3, 1, True),
    (64, 32, (128, 160), 5, 2, True), (64, 64, (64, 80), 3, 1, True),
]
# VolumeConv (networks.py:133-147): (name, kind, Cout, Cin, input (D, H, W), stride)
VOLUME = [
    ("conv0_1", "conv", 8, 64, (48, 64, 80), 1), ("conv1_0", "conv", 16, 64, (48, 64, 80), 2),
    ("conv2_0", "conv", 32, 16, (24, 32, 40), 2), ("conv3_0", "conv", 64, 32, (12, 16, 20), 2),
    ("conv3_1", "conv", 64, 64, (6, 8, 10), 1), ("conv1_1", "conv", 16, 16, (24, 32, 40), 1),
    ("conv2_1", "conv", 32, 32, (12, 16, 20), 1), ("conv4_0", "deconv", 32, 64, (6, 8, 10), 2),
    ("conv5_0", "deconv", 16, 32, (12, 16, 20), 2), ("conv6_0", "deconv", 8, 16, (24, 32, 40), 2),
    ("conv6_2", "conv", 1, 8, (48, 64, 80), 1),
]
# 1x1 convolutions over point-major rows: EdgeConv [conv1 | conv2] x 3, the flow MLP x 3 (model.py:27-43)
ROWS = [(64, 136, False), (64, 32, False), (128, 64, False), (64, 224, False), (64, 64, True), (16, 64, True)]


def _out(sp, stride):
    return tuple((s - 1) // stride + 1 for s in sp)


def _wgrad_key_conv2d(cout, cin, sp, k, stride, affine):
    o = _out(sp, stride)
    return ("conv", V, cout, cin, 1, o[0], o[1], 1, sp[0], sp[1], 1, k, k, stride, bool(affine))


def _wgrad_key_volume(kind, cout, cin, sp, stride):
    if kind == "conv":
        o = _out(sp, stride)
        if stride == 1 and cout <= 8 and cout < cin:     # operands swapped (train_ops.WGRAD_SWAP): conv0_1, conv6_2
            return ("conv", 1, cin, cout) + tuple(sp) + o + (3, 3, 3, 1, False)
        return ("conv", 1, cout, cin) + o + tuple(sp) + (3, 3, 3, stride, False)
    fine = tuple(2 * s for s in sp)                     # transposed: gr = the layer input (coarse), x = dL/dy (fine)
    return ("conv", 1, cin, cout) + tuple(sp) + fine + (3, 3, 3, 2, False)


EXPECTED_WGRAD = set(_wgrad_key_conv2d(*r) for r in TOWER) | set(_wgrad_key_volume(*r[1:]) for r in VOLUME) | \
    set(("rows", P, cg, cx, aff) for P in (P1, P2) for cg, cx, aff in ROWS)

# data gradients (train_ops helpers): (helper, dy shape, weight shape, extra)
EXPECTED_DGRAD = set()
for cout, cin, sp, k, stride, _aff in TOWER[1:]:
    EXPECTED_DGRAD.add(("conv2d_dgrad", (V, cout) + _out(sp, stride), (cout, cin, k, k), stride))
EXPECTED_DGRAD |= {
    ("conv3d_c1", (1, 1, 48, 64, 80), (1, 8, 3, 3, 3), 1),                            # conv6_2
    ("conv3d_k3_w", (1, 8, 48, 64, 80), (16, 8, 3, 3, 3), 2),                          # conv6_0 (transposed layer)
    ("conv3d_k3_w", (1, 16, 24, 32, 40), (32, 16, 3, 3, 3), 2),                        # conv5_0
    ("conv3d_bottom_w", (1, 32, 12, 16, 20), (64, 32, 3, 3, 3), 2, False),             # conv4_0
    ("conv3d_bottom_w", (1, 64, 6, 8, 10), (64, 64, 3, 3, 3), 1, True),                # conv3_1
    ("deconv3d_bottom_w", (1, 64, 6, 8, 10), (64, 32, 3, 3, 3)),                       # conv3_0
    ("conv3d_dgrad_flip", (1, 32, 12, 16, 20), (32, 32, 3, 3, 3)),                     # conv2_1
    ("conv3d_dgrad_flip", (1, 16, 24, 32, 40), (16, 16, 3, 3, 3)),                     # conv1_1
    ("conv3d_dgrad_flip", (1, 8, 48, 64, 80), (8, 64, 3, 3, 3)),                       # conv0_1
    ("deconv3d_k3s2", (1, 32, 12, 16, 20), (32, 16, 3, 3, 3)),                         # conv2_0
    ("deconv3d_k3s2", (1, 16, 24, 32, 40), (16, 64, 3, 3, 3)),                         # conv1_0
}
# BatchNorm(+ReLU) backward: planar (y shape), rows (C, points)
EXPECTED_BN = set(("planar", (V, c) + _out(sp, s)) for c, _ci, sp, _k, s, _a in TOWER[:-1]) | \
    set(("planar", (V, 64, 64, 80)) for _ in (0,)) | \
    set(("planar", (1, c) + (tuple(2 * x for x in sp) if kind == "deconv" else _out(sp, s)))
        for _n, kind, c, _ci, sp, s in VOLUME[:-1]) | \
    set(("rows", c, P) for P in (P1, P2) for c in (64, 16))


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _seeded(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _batch(dev, seed=0):
    data, img_scales, inter_scales = synthetic.make_config("cfg4", seed=seed, train_intrinsics=True)
    batch = {k: v.to(dev) for k, v in data.items()}
    batch["cam_params_list_host"] = data["cam_params_list"]
    batch["mean_host"], batch["std_host"] = data["mean"], data["std"]
    batch["gt_depth_img"] = synthetic.make_gt_depth(data, seed=seed).to(dev)
    return batch, img_scales, inter_scales


def _plan_conv(N, Cg, Cx, go, xi, k3, stride):
    plan = (ctypes.c_int * 12)()
    _lib.check(_lib.load().pf_conv_wgrad_plan(N, Cg, Cx, go[0], go[1], go[2], xi[0], xi[1], xi[2], k3[0], k3[1], k3[2],
                                              stride, plan), "pf_conv_wgrad_plan")
    return list(plan)


_PLAN_FIELDS = ("MT", "TD", "TH", "CBLK", "CBP", "NTW", "splits", "cblocks", "mblocks", "lds_bytes", "per_split", "stride")


def _report_plan(tag, plan, **errs):
    report(tag, **dict(zip(("plan_" + f for f in _PLAN_FIELDS), plan), **errs))


@pytest.mark.parametrize("Cout,Cin,sp,k,stride,affine", TOWER)
def test_tower_weight_gradient_at_cfg4_shape(dev, Cout, Cin, sp, k, stride, affine):
    """pf_conv_wgrad_f32 on the three views of a 640x512 scene, the layer input RAW with its pending per-view
    BatchNorm + ReLU where the step has one, against autograd of F.conv2d in float64."""
    x = _seeded((V, Cin) + sp, dev, 5)
    osp = _out(sp, stride)
    dy = _seeded((V, Cout) + osp, dev, 6)
    sc = sh = None
    xin = x.double()
    if affine:
        sc = (1.0 + 0.3 * _seeded((V, Cin), dev, 7)).contiguous()
        sh = (0.2 * _seeded((V, Cin), dev, 8)).contiguous()
        xin = torch.relu(xin * sc.double().view(V, Cin, 1, 1) + sh.double().view(V, Cin, 1, 1))
    w = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(xin, w, None, stride, k // 2).backward(dy.double())
    aff = None if sc is None else (sc, sh)
    dw = train_ops.conv_wgrad(dy, x, (k, k), stride, (k // 2, k // 2), aff, 1)
    assert torch.equal(dw, train_ops.conv_wgrad(dy, x, (k, k), stride, (k // 2, k // 2), aff, 1))
    err = _rel(dw, w.grad)
    plan = _plan_conv(V, Cout, Cin, (1,) + osp, (1,) + sp, (1, k, k), stride)
    _report_plan("cfg4shape_tower_wgrad_%dto%d_k%ds%d" % (Cin, Cout, k, stride), plan, rel=err)
    assert err < 2e-5, (err, plan)
    # the same shape, the same plan: what the step's launch of this layer runs with
    assert plan == _plan_conv(V, Cout, Cin, (1,) + osp, (1,) + sp, (1, k, k), stride)


@pytest.mark.parametrize("name,kind,Cout,Cin,sp,stride", VOLUME)
def test_volume_weight_gradient_at_cfg4_shape(dev, name, kind, Cout, Cin, sp, stride):
    """All eleven VolumeConv layers at 48x64x80 .. 6x8x10: Conv3d layers against autograd of F.conv3d, the three
    ConvTranspose3d layers (stride 2, padding 1, output_padding 1) against autograd of F.conv_transpose3d (result in
    nn.ConvTranspose3d's (Cin, Cout, 3, 3, 3) order), float64."""
    x = _seeded((1, Cin) + sp, dev, 9)
    if kind == "conv":
        osp = _out(sp, stride)
        dy = _seeded((1, Cout) + osp, dev, 10)
        w = torch.zeros((Cout, Cin, 3, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
        F.conv3d(x.double(), w, None, stride, 1).backward(dy.double())
        run = lambda: train_ops.conv_wgrad(dy, x, (3, 3, 3), stride, (1, 1, 1))
        plan = _plan_conv(1, Cout, Cin, osp, sp, (3, 3, 3), stride)
        if stride == 1 and Cout <= 8 and Cout < Cin:             # conv0_1, conv6_2: the launch runs with the operands swapped
            plan = _plan_conv(1, Cin, Cout, sp, osp, (3, 3, 3), 1)
    else:
        osp = tuple(2 * s for s in sp)
        dy = _seeded((1, Cout) + osp, dev, 10)
        w = torch.zeros((Cin, Cout, 3, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
        F.conv_transpose3d(x.double(), w, None, 2, 1, 1).backward(dy.double())
        run = lambda: train_ops.conv_wgrad(x, dy, (3, 3, 3), 2, (1, 1, 1))
        plan = _plan_conv(1, Cin, Cout, sp, osp, (3, 3, 3), 2)
    dw = run()
    assert torch.equal(dw, run())
    err = _rel(dw, w.grad)
    _report_plan("cfg4shape_volume_wgrad_%s" % name, plan, rel=err)
    assert dw.shape == w.shape and err < 2e-5, (err, plan)


@pytest.mark.parametrize("P", [P1, P2])
@pytest.mark.parametrize("Cg,Cx,affine", ROWS)
def test_rows_weight_gradient_at_cfg4_points(dev, P, Cg, Cx, affine):
    """pf_rows_wgrad_f32 at 25 600 / 102 400 points on strided row views (the EdgeConv layers read column slices of the
    (N, 224) concat buffer), against a float64 matrix product."""
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
    plan = (ctypes.c_int * 12)()
    _lib.check(_lib.load().pf_rows_wgrad_plan(P, Cg, Cx, plan), "pf_rows_wgrad_plan")
    err = _rel(dw, ref)
    _report_plan("cfg4shape_rows_wgrad_%dx%d_P%d" % (Cg, Cx, P), list(plan), rel=err)
    assert err < 2e-5, (err, list(plan))


@pytest.mark.parametrize("Cout,Cin,sp,k,stride,_affine", TOWER[1:])
def test_tower_data_gradient_at_cfg4_shape(dev, Cout, Cin, sp, k, stride, _affine):
    x = torch.zeros((V, Cin) + sp, dtype=torch.float64, device=dev, requires_grad=True)
    w = _seeded((Cout, Cin, k, k), dev, 15, 0.2)
    dy = _seeded((V, Cout) + _out(sp, stride), dev, 16)
    F.conv2d(x, w.double(), None, stride, k // 2).backward(dy.double())
    dx = train_ops.conv2d_dgrad(dy, w, stride)
    assert torch.equal(dx, train_ops.conv2d_dgrad(dy, w, stride))
    err = _rel(dx, x.grad)
    report("cfg4shape_tower_dgrad_%dto%d_k%ds%d" % (Cin, Cout, k, stride), rel=err)
    assert dx.shape == x.shape and err < 1e-5, err


@pytest.mark.parametrize("name,kind,Cout,Cin,sp,stride", VOLUME)
def test_volume_data_gradient_at_cfg4_shape(dev, name, kind, Cout, Cin, sp, stride):
    """The data gradient of every VolumeConv layer through the helper _VolumeTrain.backward uses for it, against
    autograd of F.conv3d / F.conv_transpose3d in float64."""
    x = torch.zeros((1, Cin) + sp, dtype=torch.float64, device=dev, requires_grad=True)
    if kind == "conv":
        w = _seeded((Cout, Cin, 3, 3, 3), dev, 17, 0.1)
        dy = _seeded((1, Cout) + _out(sp, stride), dev, 18)
        F.conv3d(x, w.double(), None, stride, 1).backward(dy.double())
    else:
        w = _seeded((Cin, Cout, 3, 3, 3), dev, 17, 0.1)
        dy = _seeded((1, Cout) + tuple(2 * s for s in sp), dev, 18)
        F.conv_transpose3d(x, w.double(), None, 2, 1, 1).backward(dy.double())
    if name == "conv6_2":
        def run():
            wf = w.flip(2, 3, 4).reshape(Cin, 27).contiguous()
            g7 = torch.empty((1, Cin) + sp, dtype=torch.float32, device=dev)
            _lib.call("pf_conv3d_k3_c1_f32", _lib.ptr(dy), _lib.ptr(wf), _lib.ptr(g7), 1, Cin, sp[0], sp[1], sp[2],
                      _lib.stream())
            return g7
    elif name in ("conv6_0", "conv5_0"):
        run = lambda: train_ops._conv3d_k3_w(dy, w, 2)
    elif name == "conv4_0":
        run = lambda: train_ops._conv3d_bottom_w(dy, w, 2)
    elif name == "conv3_1":
        run = lambda: train_ops._conv3d_bottom_w(dy, w, 1, flip_t=True)
    elif name == "conv3_0":
        run = lambda: train_ops._deconv3d_bottom_w(dy, w)
    elif name in ("conv2_0", "conv1_0"):
        run = lambda: pointflow.deconv3d_k3s2(dy, None, w, False)[0]
    else:
        run = lambda: train_ops.conv3d_dgrad_flip(dy, w)
    with pointflow.no_pack_cache():
        dx = run()
        assert torch.equal(dx, run())
    err = _rel(dx, x.grad)
    report("cfg4shape_volume_dgrad_%s" % name, rel=err)
    assert dx.shape == x.shape and err < 1e-5, err


_BN_PLANAR = sorted(set(k[1] for k in EXPECTED_BN if k[0] == "planar"))


@pytest.mark.parametrize("shape", _BN_PLANAR, ids=lambda s: "x".join(str(v) for v in s))
def test_bn_relu_backward_at_cfg4_shape(dev, shape, monkeypatch):
    """The BatchNorm backward on every (samples, channels, spatial) the step has -- the towers' per-view statistics (3
    statistic groups), VolumeConv's single sample -- in the form the step runs it (pf_bn_bwd_plane_f32 where a plane fits
    one block, else pf_bn_bwd_reduce + pf_bn_bwd_apply_fused)."""
    import test_gpu_train_ops as T
    T.test_bn_relu_backward_vs_float64_autograd(dev, shape[0], shape[1], tuple(shape[2:]), 1, True, 1, monkeypatch)


@pytest.mark.parametrize("P", [P1, P2])
@pytest.mark.parametrize("C", [64, 16])
def test_rows_bn_relu_backward_at_cfg4_points(dev, P, C):
    """The flow MLP's BatchNorm1d + ReLU backward on point-major rows (pf_rows_bn_bwd_reduce / pf_bn_bwd_coeffs /
    pf_rows_bn_bwd_apply) against autograd of F.batch_norm(training) + relu in float64."""
    y = _seeded((P, C), dev, 21, 2.0) + 0.3
    g = _seeded((P, C), dev, 22)
    bn = torch.nn.BatchNorm1d(C).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(1.0 + 0.2 * _seeded((C,), dev, 23))
        bn.bias.copy_(0.1 * _seeded((C,), dev, 24))
    yd = y.double()
    parts = torch.stack([yd.sum(0), (yd * yd).sum(0)], dim=1).view(1, 1, C, 2).contiguous()
    rows = train_ops.bn_train_rows(bn, parts, 0, C, float(P), 1, 1)
    dy, dgamma, dbeta = train_ops.rows_bn_backward(g, y, rows, C, 1, P, 1, True)
    dy2, dgamma2, dbeta2 = train_ops.rows_bn_backward(g, y, rows, C, 1, P, 1, True)
    assert torch.equal(dy, dy2) and torch.equal(dgamma, dgamma2) and torch.equal(dbeta, dbeta2)
    yr = yd.clone().requires_grad_(True)
    wr, br = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    torch.relu(F.batch_norm(yr, None, None, wr, br, True, 0.0, bn.eps)).backward(g.double())
    e = dict(dy=_rel(dy, yr.grad), dgamma=_rel(dgamma, wr.grad), dbeta=_rel(dbeta, br.grad))
    report("cfg4shape_rows_bn_bwd_C%d_P%d" % (C, P), **e)
    assert e["dy"] < 1e-5 and e["dgamma"] < 2e-5 and e["dbeta"] < 2e-5, e


def test_image_tower_node_at_cfg4_size_vs_float64(dev):
    """The tower node on the three 512x640 views of a cfg-4 scene against the float64 ATen modules: outputs at 2e-5,
    parameter gradients within max(2e-4, 3 x what the ORACLE's float32 evaluation of the same tower deviates by from float64
    on the same inputs) -- at this size a float32 backward through eleven BatchNorm + ReLU layers is 1e-3 .. 1e-2 from float64
    per tensor whoever computes it (tests/test_gpu_train_ops.py::_oracle32_yardstick)."""
    import test_gpu_train_ops as T
    T.test_image_tower_node_vs_float64_autograd(dev, (H, W), yardstick=True)


def test_volume_conv_node_at_cfg4_size_vs_float64(dev):
    """The VolumeConv node on a (1, 64, 48, 64, 80) cost volume against the float64 ATen module."""
    import test_gpu_train_ops as T
    T.test_volume_conv_node_vs_float64_autograd(dev, (D, H // 8, W // 8), yardstick=True)


def test_train_step_gradient_is_bit_reproducible_at_cfg4(dev):
    import test_gpu_model as TM
    TM.test_train_step_gradient_is_bit_reproducible(dev, "cfg4")


def test_graphed_train_step_matches_the_eager_step_at_cfg4(dev):
    import test_gpu_model as TM
    TM.test_graphed_train_step_matches_the_eager_step(dev, "cfg4")


def test_step_launches_exactly_the_tabulated_shapes(dev, monkeypatch):
    """One real cfg-4 step, recorded: the weight-gradient launches (C ABI arguments), the data-gradient helpers and the
    BatchNorm-backward helpers see exactly the shapes of the tables above -- what the per-shape tests below cover."""
    from pointmvsnet_amd.model import PointMVSNet
    from pointmvsnet_amd.train_step import TrainStep
    seen_w, seen_d, seen_bn = set(), set(), set()
    real_call = _lib.call

    def call(name, *args, **kw):
        if name == "pf_conv_wgrad_f32":
            seen_w.add(("conv",) + tuple(int(a) for a in args[3:16]) + (args[19] is not None,))
        elif name == "pf_conv_wgrad_batch_f32":              # a node's layers, queued and issued together
            for it in args[0][:int(args[1])]:
                if int(it.rows_P) > 0:                           # a 1x1 layer on point-major rows, deferred to the end
                    seen_w.add(("rows", int(it.rows_P), int(it.Cg), int(it.Cx), bool(it.x_scale)))
                    continue
                seen_w.add(("conv", int(it.N), int(it.Cg), int(it.Cx), int(it.Do), int(it.Ho), int(it.Wo), int(it.Di),
                            int(it.Hi), int(it.Wi), int(it.KD), int(it.KH), int(it.KW), int(it.stride),
                            bool(it.x_scale)))
        elif name == "pf_rows_wgrad_f32":
            seen_w.add(("rows", int(args[5]), int(args[6]), int(args[7]), args[8] is not None))
        elif name == "pf_conv3d_k3_c1_f32":
            seen_d.add(("conv3d_c1", (1, 1) + tuple(int(a) for a in args[5:8]), (1, int(args[4]), 3, 3, 3), 1))
        return real_call(name, *args, **kw)

    monkeypatch.setattr(_lib, "call", call)

    def wrap(mod, fname, key):
        real = getattr(mod, fname)

        def inner(*a, **kw):
            k = key(*a, **kw)
            if k is not None:
                (seen_bn if k[0] in ("planar", "rows") else seen_d).add(k)
            return real(*a, **kw)
        monkeypatch.setattr(mod, fname, inner)

    wrap(train_ops, "conv2d_dgrad", lambda dy, w, s: ("conv2d_dgrad", tuple(dy.shape), tuple(w.shape), int(s)))
    wrap(train_ops, "conv3d_dgrad_flip", lambda dy, w: ("conv3d_dgrad_flip", tuple(dy.shape), tuple(w.shape)))
    wrap(train_ops, "_conv3d_k3_w", lambda x, w, s: ("conv3d_k3_w", tuple(x.shape), tuple(w.shape), int(s)))
    wrap(train_ops, "_conv3d_bottom_w",
         lambda x, w, s, flip_t=False: ("conv3d_bottom_w", tuple(x.shape), tuple(w.shape), int(s), bool(flip_t)))
    wrap(train_ops, "_deconv3d_bottom_w", lambda x, w: ("deconv3d_bottom_w", tuple(x.shape), tuple(w.shape)))
    wrap(pointflow, "deconv3d_k3s2",
         lambda x, skip, w, stats, *a, **kw: None if stats else ("deconv3d_k3s2", tuple(x.shape), tuple(w.shape)))
    wrap(train_ops, "bn_backward", lambda g, y, rows, sps, relu=True, into=None: ("planar", tuple(y.shape)))
    wrap(train_ops, "rows_bn_backward",
         lambda g, y, rows, C, G, Ng, gps, relu=True, into=None: ("rows", int(C), int(G) * int(Ng)))

    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    batch, img_scales, inter_scales = _batch(dev)
    loss, _, _ = TrainStep(net)(batch, img_scales, inter_scales)
    assert torch.isfinite(loss)
    assert seen_w == EXPECTED_WGRAD, (sorted(seen_w - EXPECTED_WGRAD), sorted(EXPECTED_WGRAD - seen_w))
    assert seen_d == EXPECTED_DGRAD, (sorted(seen_d - EXPECTED_DGRAD, key=str), sorted(EXPECTED_DGRAD - seen_d, key=str))
    assert seen_bn == EXPECTED_BN, (sorted(seen_bn - EXPECTED_BN, key=str), sorted(EXPECTED_BN - seen_bn, key=str))


def test_train_step_parameter_gradients_vs_oracle_autograd_at_cfg4(dev, monkeypatch):
    """The whole 698 936-element gradient of ONE cfg-4 step against the oracle's autograd in float32 and float64 on the
    host, the oracle's kNN injected on both sides: the gates of the "tiny" arm (tests/test_gpu_model.py)."""
    import test_gpu_model as TM
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))        # (one thread per logical CPU of the GPU box runs the oracle 4x slower)
    try:
        TM.test_train_step_parameter_gradients_vs_oracle_autograd(dev, monkeypatch, "cfg4", 1)
    finally:
        torch.set_num_threads(threads)


# ---------------------------------------------------------------------------------------------
# float64 arms of the node tests (round 4 compared these nodes with compositions of this package's own operators)
# ---------------------------------------------------------------------------------------------
def _grads(params):
    return [p.grad.detach().clone() for p in params]


def _tiny_plan(dev):
    from pointmvsnet_amd.model import PointMVSNet
    data, img_scales, inter_scales = synthetic.make_config("tiny", train_intrinsics=True)
    batch = {k: v.to(dev) for k, v in data.items()}
    batch["cam_params_list_host"] = data["cam_params_list"]
    batch["mean_host"], batch["std_host"] = data["mean"], data["std"]
    net = PointMVSNet().to(dev).train()
    tplan = net.make_train_plan(batch, img_scales, inter_scales, isTest=False)
    torch.cuda.synchronize()
    return net, tplan, data["img_list"].shape[1], data, img_scales, inter_scales


class _float64_on(object):
    """The oracle's functions (oracle/pointflow_oracle.py: the reference's formulas on plain ATen operators) evaluated
    in float64 on the GPU: factory calls inside follow the default dtype / device."""

    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        torch.set_default_dtype(torch.float64)
        self.ctx = torch.device(self.dev)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        torch.set_default_dtype(torch.float32)
        return False


@pytest.mark.parametrize("it,h,w", [(0, 16, 24), (1, 32, 48)])
def test_flow_feature_node_vs_float64_oracle(dev, it, h, w):
    """Feature assembly of a PointFlow iteration (resize + warp + variance + xyz; reference model.py:153-204) as one
    node -- backward on csrc/warp_bwd.hip, no atomics -- against the ORACLE's flow_point_features (F.interpolate,
    grid_sample with the reference's grid normalisation, variance over views) evaluated in float64 on the same device
    from the scene's cameras: the feature rows, the gradients w.r.t. the three pyramid levels and w.r.t. the prior
    depth; and bit-reproducibility.  Nothing of this package runs on the reference side."""
    from oracle import pointflow_oracle as O
    net, tplan, V, data, img_scales, inter_scales = _tiny_plan(dev)
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
    # float64: the oracle's formulas from the scene's own cameras (train intrinsics are given at 1/4 resolution)
    with _float64_on(dev):
        cams = data["cam_params_list"].to(dev).double()
        ext, _R, t, R_inv = O.split_cameras(cams)
        Kf = cams[:, :, 1, :3, :3].clone()
        Kf[:, :, :2, :3] *= 4 * img_scales[it]
        interval = inter_scales[it] * cams[:, 0, 1, 3, 1]
        pyr64 = {n: pyr[n].detach().double().requires_grad_(True) for n in pyr}
        depth64 = depth.detach().double().requires_grad_(True)
        feature, xyz_ref = O.flow_point_features(pyr64, depth64, interval, Kf, ext, R_inv, t,
                                                 data["mean"].to(dev).double(), data["std"].to(dev).double())
        rows_ref = feature.view(136, 5 * h * w).t()
        (rows_ref * gfeat.double()).sum().backward()
    scale = float(rows_ref.abs().max())
    e = dict(rows=float((runs[0][0].double() - rows_ref).abs().max()) / scale,
             xyz=_rel(runs[0][1].view(-1), xyz_ref.reshape(-1)), ddepth=_rel(runs[0][2], depth64.grad))
    for i, n in enumerate(pyr):
        e["d" + n] = _rel(runs[0][3 + i], pyr64[n].grad)
    report("flow_feature_node_f64_it%d" % it, **e)
    # the kernels project in float32, the reference side in float64: a tap position differs by ~1e-5 texel on maps with
    # O(1) texel contrast, which is the error floor of the rows and of the level gradients here
    assert e["rows"] < 1e-4 and e["xyz"] < 1e-5, e
    assert e["ddepth"] < 1e-4 and max(e["dconv1"], e["dconv2"], e["dconv3"]) < 2e-4, e


def test_coarse_volume_node_vs_float64_oracle(dev):
    """Coarse cost volume (reference model.py:79-111) as one node against the ORACLE's composition (frustum by matmul,
    fetch_features = grid_sample, the reference view's un-warped map, variance over views) in float64 on the same
    device from the scene's cameras: the volume, the frustum points, the gradient w.r.t. the tower maps."""
    from oracle import pointflow_oracle as O
    net, tplan, V, data, img_scales, inter_scales = _tiny_plan(dev)
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
    with _float64_on(dev):
        cams = data["cam_params_list"].to(dev).double()
        ext, _R, t, R_inv = O.split_cameras(cams)
        K = cams[:, :, 1, :3, :3].clone()
        K[:, :, :2, :3] = K[:, :, :2, :3] / 2.0                               # oracle forward(): train mode
        d_start, d_int = cams[:, 0, 1, 3, 0], cams[:, 0, 1, 3, 1]
        d_end = d_start + (D - 1) * d_int
        depths = torch.linspace(float(d_start[0]), float(d_end[0]), D).view(1, 1, 1, D, 1)
        grid = O.pixel_grid(FH, FW).view(1, 1, 3, -1)
        uv = torch.matmul(torch.inverse(K[:, 0]).unsqueeze(1), grid)
        cam_pts = (uv.unsqueeze(3) * depths).view(1, 1, 3, -1)
        world_ref = torch.matmul(R_inv[:, 0:1], cam_pts - t[:, 0:1]).transpose(1, 2).contiguous().view(1, 3, -1)
        m64 = maps.detach().double().requires_grad_(True)
        fl = m64.unsqueeze(0)
        pf = O.fetch_features(fl, world_ref, K, ext)
        ref0 = fl[:, 0].unsqueeze(2).expand(-1, -1, D, -1, -1).contiguous().view(1, C, -1)
        pf = torch.cat([ref0.unsqueeze(1), pf[:, 1:]], dim=1)
        cost_ref = O.variance_over_views(pf)
        (cost_ref * gcost.double()).sum().backward()
    e = dict(cost=float((runs[0][0].double() - cost_ref).abs().max()) / float(cost_ref.abs().max()),
             world=_rel(runs[0][1], world_ref), dmaps=_rel(runs[0][2], m64.grad))
    report("coarse_volume_node_f64", **e)
    assert e["cost"] < 1e-4 and e["world"] < 1e-6 and e["dmaps"] < 2e-4, e       # (float32 projection, as above)


def test_edge_chain_and_mlp_nodes_vs_float64_functional(dev):
    """EdgeConv x3 + SharedMLP on point-major rows (two nodes) against a float64 composition of plain ATen operators on
    the same weights (F.conv1d, an explicit torch.gather of the neighbours, F.batch_norm with batch statistics, relu,
    mean over k: reference networks.py:18-45,56-81 CUDA branch, nn/mlp.py:45-81): outputs, input gradient, parameter
    gradients.  Nothing of this package runs on the reference side (the neighbour indices are an input of both)."""
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
    # float64, functional: leaves in the order of ``params``
    leaves = [p.detach().double().requires_grad_(True) for p in params]
    it = iter(leaves)
    fd = feat.detach().double().requires_grad_(True)
    x = fd.t().unsqueeze(0)
    k = idx.shape[2]
    outs = []
    for m in net.flow_edge_conv:
        w1, w2, gamma, beta = next(it), next(it), next(it), next(it)
        l, e = F.conv1d(x, w1), F.conv1d(x, w2)
        nb = torch.gather(e.unsqueeze(3).expand(-1, -1, -1, k), 2, idx.unsqueeze(1).expand(-1, e.shape[1], -1, -1))
        central = l.unsqueeze(-1).expand(-1, -1, -1, k)
        edge = torch.cat([central, nb - central], dim=1) if m.concat else nb - central
        x = torch.relu(F.batch_norm(edge, None, None, gamma, beta, True, 0.0, m.bn.eps)).mean(dim=3)
        outs.append(x)
    y = torch.cat(outs, dim=1)
    for blk in net.flow_mlp[0]:
        wc, gamma, beta = next(it), next(it), next(it)
        y = torch.relu(F.batch_norm(F.conv1d(y, wc), None, None, gamma, beta, True, 0.0, blk.bn.eps))
    ract = y[0].t()
    (ract * g.double()).sum().backward()
    errs = sorted(((_rel(a, p.grad), i) for i, (a, p) in enumerate(zip(mine, leaves))), reverse=True)
    e_act, e_x = _rel(act, ract), _rel(gfeat, fd.grad)
    report("edge_chain_mlp_nodes_functional", act_rel=e_act, dfeature_rel=e_x, worst_grad_rel=errs[0][0],
           median_grad_rel=errs[len(errs) // 2][0])
    # the forward agrees to float32 rounding; in the backward a float32 and a float64 evaluation legitimately disagree
    # about the ReLU mask of the pre-activations within rounding of zero (a few per thousand of the 16 N edge values,
    # tests/test_gpu_backward_cfg4.py), which moves single gradient entries: measured 6e-4 / 4e-3 / median 3e-4
    assert e_act < 2e-5 and e_x < 3e-3 and errs[0][0] < 1e-2 and errs[len(errs) // 2][0] < 1e-3, (e_act, e_x, errs[:5])



def _tiny_step_gradient(dev):
    from pointmvsnet_amd.model import PointMVSNet
    from pointmvsnet_amd.train_step import TrainStep
    data, img_scales, inter_scales = synthetic.make_config("tiny", train_intrinsics=True)
    batch = {k: v.to(dev) for k, v in data.items()}
    batch["cam_params_list_host"] = data["cam_params_list"]
    batch["mean_host"], batch["std_host"] = data["mean"], data["std"]
    batch["gt_depth_img"] = synthetic.make_gt_depth(data).to(dev)
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    step = TrainStep(net)
    loss, _, _ = step(batch, img_scales, inter_scales)
    buffers = torch.cat([b.detach().reshape(-1).double() for b in net.buffers()])
    return float(loss), step.bucket.flat.detach().clone(), buffers


def test_lazy_bn_step_equals_the_eager_finalize_step(dev, monkeypatch):
    """PF_TRAIN_LAZY_BN (default 1 since round 6: the training forward's BatchNorms resolved by their consumers, the rows
    for the backward from ONE batched finalize at the end of the forward; VolumeConv's normalise passes finalize for
    themselves) against PF_TRAIN_LAZY_BN=0 (one pf_bn_train_rows_f32 launch per BatchNorm): the same loss and gradient to
    float32 rounding of a statistic's summation order, the same running statistics, and bit-reproducible.  (The oracle
    gates and graphed == eager run with the default, i.e. with it on: tests/test_gpu_model.py and the cfg-4 tests above.)"""
    monkeypatch.setattr(train_ops, "TRAIN_LAZY_BN", 0)
    l0, g0, b0 = _tiny_step_gradient(dev)
    monkeypatch.setattr(train_ops, "TRAIN_LAZY_BN", 1)
    l1, g1, b1 = _tiny_step_gradient(dev)
    l2, g2, b2 = _tiny_step_gradient(dev)
    assert l1 == l2 and torch.equal(g1, g2) and torch.equal(b1, b2)
    e = dict(loss=abs(l1 - l0) / abs(l0), grad_l2=float((g1 - g0).norm() / g0.norm()),
             buffers=float((b1 - b0).abs().max() / b0.abs().max()))
    report("lazy_bn_vs_eager_finalize", **e)
    assert e["loss"] < 1e-6 and e["grad_l2"] < 1e-4 and e["buffers"] < 1e-5, e
