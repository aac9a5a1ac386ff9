"""GPU parity tests, operator level: every HIP entry point against the CPU oracle on identical
seeded inputs and against the golden vectors produced by the reference itself.

Tolerances (float ops) are stated where used; integer / index work is bit-exact.  Measured errors are
appended to gpurun_out/parity_report.jsonl.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, report
from oracle import bruteforce as BF
from oracle import pointflow_oracle as O
from pointmvsnet_amd import _lib, pointflow, synthetic
from pointmvsnet_amd.functions.gather_knn import GatherKNN, dgcnn_ext, gather_knn
from pointmvsnet_amd.networks import EdgeConv, EdgeConvNoC, VolumeConv
from pointmvsnet_amd.utils.feature_fetcher import FeatureFetcher, fetch_variance
from pointmvsnet_amd.utils.torch_utils import get_knn_3d, knn_lattice

pytestmark = pytest.mark.gpu


def _maxabs(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


# ---------------------------------------------------------------------------------------------
# row G
# ---------------------------------------------------------------------------------------------
def test_gather_knn_reference_selftest(dev):
    # the reference's own known-answer test (functions/gather_knn.py:27-56), forward and backward
    g = load_golden("gather_knn_selftest")
    f = g["feature"].to(dev).requires_grad_(True)
    out = gather_knn(f, g["index"].to(dev))
    assert torch.equal(out.cpu(), g["out"])
    out.backward(torch.ones_like(out))
    assert torch.allclose(f.grad.cpu(), g["grad"], rtol=1e-6, atol=1e-6)
    assert _lib.status() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,C,N,K", [(1, 1, 1, 1), (2, 7, 133, 5), (1, 64, 5000, 16), (3, 32, 257, 1)])
def test_gather_knn_forward_backward(dev, dtype, B, C, N, K):
    g = torch.Generator().manual_seed(B * 1000 + N)
    x = torch.randn(B, C, N, generator=g, dtype=dtype)
    idx = torch.randint(0, N, (B, N, K), generator=g)
    out = dgcnn_ext.gather_knn_forward(x.to(dev), idx.to(dev))
    assert torch.equal(out.cpu(), O.gather_knn(x, idx))                      # bit-exact copy
    go = torch.randn(B, C, N, K, generator=g, dtype=dtype)
    gi = dgcnn_ext.gather_knn_backward(go.to(dev), idx.to(dev))
    ref = O.gather_knn_backward(go.double(), idx)
    err = _maxabs(gi, ref)
    report("gather_bwd_%s" % str(dtype), err=err)
    # atomics reorder the (at most N*K) additions: float32 1e-5 abs on O(sqrt(K)) sums, float64 1e-12
    assert err < (2e-5 if dtype == torch.float32 else 1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gather_knn_backward_is_bit_reproducible_and_drops_bad_indices(dev, dtype, monkeypatch):
    """The default backward gathers over the inverted index lists: identical bits from run to run (many slots per
    target here: N = 300 points named by 300 * 16 slots drawn from 40 targets), out-of-range slots carry no gradient
    like the forward's zeros, and the reference's atomic scatter agrees to rounding."""
    B, C, N, K = 2, 9, 300, 16
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, 40, (B, N, K), generator=g)
    idx[0, 5, 3], idx[1, 7, 0] = -1, N                                       # flagged; no gradient through them
    go = torch.randn(B, C, N, K, generator=g, dtype=dtype)
    first = dgcnn_ext.gather_knn_backward(go.to(dev), idx.to(dev))
    assert _lib.status() & 1
    second = dgcnn_ext.gather_knn_backward(go.to(dev), idx.to(dev))
    assert _lib.status() & 1 and torch.equal(first, second)
    valid = (idx >= 0) & (idx < N)
    ref = O.gather_knn_backward((go.double() * valid.unsqueeze(1)), idx.clamp(0, N - 1))
    assert _maxabs(first, ref) < (1e-4 if dtype == torch.float32 else 1e-11)
    monkeypatch.setattr(pointflow, "DETERMINISTIC_BACKWARD", False)
    scattered = dgcnn_ext.gather_knn_backward(go.to(dev), idx.to(dev))
    assert _lib.status() & 1
    assert _maxabs(first, scattered) < (1e-4 if dtype == torch.float32 else 1e-11)


def test_gather_knn_empty_and_noncontiguous(dev):
    x = torch.randn(2, 4, 9, device=dev)
    idx = torch.randint(0, 9, (2, 9, 0), device=dev)
    assert gather_knn(x, idx).shape == (2, 4, 9, 0)
    xt = torch.randn(2, 9, 4, device=dev).transpose(1, 2)                    # non-contiguous input
    idx = torch.randint(0, 9, (2, 9, 3), device=dev)
    assert torch.equal(gather_knn(xt, idx).cpu(), O.gather_knn(xt.cpu().contiguous(), idx.cpu()))


def test_gather_knn_errors(dev):
    x = torch.randn(2, 4, 9, device=dev)
    with pytest.raises(RuntimeError):
        gather_knn(x.cpu(), torch.zeros(2, 9, 3, dtype=torch.int64))       # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        gather_knn(x, torch.zeros(2, 8, 3, dtype=torch.int64, device=dev))  # shape mismatch
    with pytest.raises(RuntimeError):
        gather_knn(x, torch.zeros(2, 9, 3, dtype=torch.int32, device=dev))  # index dtype
    bad = torch.full((2, 9, 3), 9, dtype=torch.int64, device=dev)           # out of range: flagged, no fault
    out = gather_knn(x, bad)
    assert _lib.status() & 1
    assert float(out.abs().max()) == 0.0
    assert _lib.status() == 0


# ---------------------------------------------------------------------------------------------
# row K
# ---------------------------------------------------------------------------------------------
def _check_knn_against_reference_idx(xyz_cpu, idx_gpu, ref_idx, ks, knn):
    """Index SETS must match the reference; rows may differ only inside exact-tie groups."""
    mine = np.sort(idx_gpu.cpu().numpy(), axis=2)
    ref = np.sort(ref_idx.numpy(), axis=2)
    n_diff = 0
    for b in range(mine.shape[0]):
        rows = np.where((mine[b] != ref[b]).any(axis=1))[0]
        if len(rows) == 0:
            continue
        d2 = BF.knn_window_d2(xyz_cpu[b].numpy(), ks)
        srt = np.sort(d2, axis=0)
        for n in rows:                    # a tie at the k-th rank is the only legal reason
            assert knn < srt.shape[0] and srt[knn - 1, n] == srt[knn, n]
        n_diff += len(rows)
    return n_diff


@pytest.mark.parametrize("name,ks,knn", [("knn_lattice_far", 5, 16), ("knn_lattice_origin", 5, 16),
                                         ("knn_lattice_k3", 3, 8)])
def test_knn_lattice_vs_reference_and_tie_rule(dev, name, ks, knn):
    g = load_golden(name)
    idx, codes = knn_lattice(g["xyz"].to(dev), ks, knn, with_codes=True)
    assert idx.dtype == torch.int64 and idx.shape == g["idx"].shape
    for b in range(g["xyz"].shape[0]):
        bf_idx, bf_code = BF.knn_window(g["xyz"][b].numpy(), ks, knn)
        assert np.array_equal(idx[b].cpu().numpy(), bf_idx)                  # bit-exact incl. order (stated tie rule)
        assert np.array_equal(codes[b].cpu().numpy(), bf_code)
    n_diff = _check_knn_against_reference_idx(g["xyz"], idx, g["idx"], ks, knn)
    report("knn_rows_in_tie_groups_" + name, rows=n_diff)
    if name == "knn_lattice_far":
        assert n_diff == 0


def test_knn_lattice_strided_view_and_api(dev):
    g = load_golden("knn_lattice_strided")
    full = g["xyz_full"].to(dev)
    sub = full.view(1, 3, 5, 6, 2, 8, 2)[:, :, :, :, 1, :, 0]               # as model.py:251-252
    assert not sub.is_contiguous()
    idx = get_knn_3d(sub, 5, knn=16)
    assert _check_knn_against_reference_idx(sub.cpu().contiguous(), idx, g["idx"], 5, 16) == 0
    assert get_knn_3d(sub, 5).shape == (1, 5 * 6 * 8, 20)                    # default knn=20
    with pytest.raises(AssertionError):
        get_knn_3d(sub, 4, knn=8)                                            # even window (reference asserts)


@pytest.mark.parametrize("shape", [(1, 5, 1, 1), (2, 5, 3, 70), (1, 1, 9, 33), (1, 5, 64, 80)])
def test_knn_lattice_shapes_vs_oracle(dev, shape):
    B, D, H, W = shape
    g = torch.Generator().manual_seed(D * H * W)
    xyz = torch.randn(B, 3, D, H, W, generator=g) * 0.3 + torch.tensor([1.0, 2.0, -3.0]).view(1, 3, 1, 1, 1)
    knn = min(16, 125)
    idx = knn_lattice(xyz.to(dev), 5, knn)
    for b in range(B):
        bf_idx, _ = BF.knn_window(xyz[b].numpy(), 5, knn)
        assert np.array_equal(idx[b].cpu().numpy(), bf_idx)
    assert _check_knn_against_reference_idx(xyz, idx, O.knn_lattice(xyz, 5, knn), 5, knn) >= 0


def test_knn_full_size_properties(dev):
    # BASELINE config-2 flow-2 size (4 sub-grids of 5x64x80): properties that need no oracle
    g = torch.Generator().manual_seed(5)
    G, D, H, W = 4, 5, 64, 80
    base = torch.stack(torch.meshgrid(torch.arange(W).float(), torch.arange(H).float(), torch.arange(D).float(),
                                      indexing="xy"), 0)                    # (3,H,W,D)
    xyz = base.permute(0, 3, 1, 2).unsqueeze(0).repeat(G, 1, 1, 1, 1) * 0.1 + 5.0
    xyz = xyz + 0.01 * torch.randn(xyz.shape, generator=g)
    idx, codes = knn_lattice(xyz.to(dev), 5, 16, with_codes=True)
    N = D * H * W
    idx_c = idx.cpu()
    assert torch.equal(idx_c[:, :, 0], torch.arange(N).view(1, N).expand(G, N))   # self is the nearest
    assert int(codes[:, :, 0].min()) == 62 and int(codes[:, :, 0].max()) == 62      # centre code
    assert int(idx_c.min()) >= 0 and int(idx_c.max()) < N
    assert (idx_c.sort(dim=2)[0].diff(dim=2) > 0).all()                             # 16 distinct neighbours
    again = knn_lattice(xyz.to(dev), 5, 16)
    assert torch.equal(again, idx)                                                   # deterministic


# ---------------------------------------------------------------------------------------------
# rows W, V
# ---------------------------------------------------------------------------------------------
def test_fetch_reference_selftest(dev):
    g = load_golden("feature_fetch_selftest")
    torch.manual_seed(0)
    _ = torch.rand(3, 2, 3, 4)
    feats = torch.rand(3, 2, 16, 240, 320)
    out = FeatureFetcher()(feats.to(dev), g["pts"].to(dev), g["K"].to(dev), g["E"].to(dev))
    err = _maxabs(out, g["out"])
    report("fetch_selftest", err=err)
    # the reference's own acceptance (utils/feature_fetcher.py:97): rtol 1e-2 against the texel
    assert np.allclose(out[:, 0, :, 0].cpu().numpy(), feats[:, 0, :, 80, 60].numpy(), rtol=1e-2, atol=1e-4)
    assert err < 2e-3        # ill-conditioned random extrinsics amplify 1-ulp projection differences


def test_fetch_random_vs_reference(dev):
    g = load_golden("feature_fetch_random")
    fetcher = FeatureFetcher()
    out = fetcher(g["feats"].to(dev), g["pts"].to(dev), g["K"].to(dev), g["E"].to(dev))
    scale = float(g["feats"].abs().max())
    err = _maxabs(out, g["out"])
    report("fetch_random", err=err, scale=scale)
    # float32 projection rounds to ~1 ulp of the pixel coordinate (4e-6 at u~50): tolerance 3e-5 * max|map|
    assert err < 3e-5 * scale
    assert out.is_contiguous() and out.shape == g["out"].shape
    out[:, 0] = 1.0                                                          # writable, like model.py:106
    # identity extrinsics path (cam_extrinsics=None, feature_fetcher.py:33-35)
    o2 = fetcher(g["feats"].to(dev), g["pts"].to(dev), g["K"].to(dev), None)
    r2 = O.fetch_features(g["feats"], g["pts"], g["K"], None)
    assert _maxabs(o2, r2) < 3e-5 * scale


def test_fetch_backward_vs_autograd_oracle(dev):
    g = load_golden("feature_fetch_random")
    feats = g["feats"].clone().requires_grad_(True)
    ref = O.fetch_features(feats, g["pts"], g["K"], g["E"])
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    ref.backward(go)
    f2 = g["feats"].to(dev).requires_grad_(True)
    out = FeatureFetcher()(f2, g["pts"].to(dev), g["K"].to(dev), g["E"].to(dev))
    out.backward(go.to(dev))
    err = _maxabs(f2.grad, feats.grad)
    report("fetch_backward", err=err, scale=float(feats.grad.abs().max()))
    assert err < 1e-4 * float(feats.grad.abs().max())


@pytest.mark.parametrize("V", [2, 3, 5, 7])
@pytest.mark.parametrize("ref_override", [False, True])
def test_fetch_variance_vs_oracle(dev, V, ref_override):
    gen = torch.Generator().manual_seed(V)
    B, C, H, W, D = 1, 8, 16, 20, 3
    data = synthetic.make_scene(128, 160, V, D, seed=V, batch=B)
    cams = data["cam_params_list"]
    K = cams[:, :, 1, :3, :3].clone()
    K[:, :, :2, :3] /= 8.0
    E = cams[:, :, 0, :3, :4].clone()
    feats = torch.randn(B, V, C, H, W, generator=gen)
    N = D * H * W
    pts = torch.randn(B, 3, N, generator=gen) * torch.tensor([40.0, 30.0, 80.0]).view(1, 3, 1) \
        + torch.tensor([0.0, 0.0, 600.0]).view(1, 3, 1)
    pf = O.fetch_features(feats, pts, K, E)
    if ref_override:
        pf[:, 0] = feats[:, 0].unsqueeze(2).expand(-1, -1, D, -1, -1).contiguous().view(B, C, -1)
    ref = O.variance_over_views(pf)
    out = fetch_variance(feats.to(dev), pts.to(dev), K.to(dev), E.to(dev), ref_override=ref_override)
    scale = float((pf ** 2).max())
    err = _maxabs(out, ref)
    report("fetch_variance_V%d_%d" % (V, int(ref_override)), err=err, scale=scale)
    # E[x^2]-E[x]^2 cancels: absolute accuracy is that of E[x^2] -> 6e-5 * max|x|^2
    assert err < 6e-5 * scale


@pytest.mark.parametrize("B,V", [(1, 3), (2, 5)])
def test_frustum_variance_vs_reference_composition(dev, B, V):
    """Coarse-stage use (model.py:79-111): frustum points generated in the kernel == the reference's matmul
    composition to float32 rounding, and the cost volume == fetch_variance on exactly those points."""
    from pointmvsnet_amd.functions.functions import get_pixel_grids
    from pointmvsnet_amd.utils.feature_fetcher import frustum_variance
    gen = torch.Generator().manual_seed(10 * B + V)
    C, H, W, D = 8, 16, 20, 6
    data = synthetic.make_scene(128, 160, V, D, seed=V, batch=B)
    cams = data["cam_params_list"]
    K = cams[:, :, 1, :3, :3].clone()
    K[:, :, :2, :3] /= 8.0
    E = cams[:, :, 0, :3, :4].clone()
    feats = torch.randn(B, V, C, H, W, generator=gen)
    kinv = torch.inverse(K[:, 0])
    rinv = torch.inverse(E[:, 0, :, :3])
    t0 = E[:, 0, :, 3]
    depths = torch.stack([torch.linspace(425.0 + 10 * b, 900.0 + 10 * b, D) for b in range(B)])
    grid = get_pixel_grids(H, W).view(1, 3, -1).expand(B, 3, -1)
    uv = torch.matmul(kinv, grid)
    cam_points = (uv.unsqueeze(2) * depths.view(B, 1, D, 1)).view(B, 3, -1)
    world_ref = torch.matmul(rinv, cam_points - t0.unsqueeze(2))
    out, world = frustum_variance(feats.to(dev), kinv.to(dev), rinv.to(dev), t0.to(dev), depths.to(dev), K.to(dev),
                                  E.to(dev))
    err = _maxabs(world, world_ref)
    report("frustum_points_B%d_V%d" % (B, V), err=err, scale=float(world_ref.abs().max()))
    # float32 evaluation of a ~1000 mm coordinate in a different association order: a few ulp (6e-5 mm each)
    assert err < 1e-3
    same = fetch_variance(feats.to(dev), world, K.to(dev), E.to(dev), ref_override=True)
    assert torch.equal(out, same)
    out2, none = frustum_variance(feats.to(dev), kinv.to(dev), rinv.to(dev), t0.to(dev), depths.to(dev), K.to(dev),
                                  E.to(dev), want_points=False)
    assert none is None and torch.equal(out2, out)


def test_fetch_errors(dev):
    f = FeatureFetcher()
    with pytest.raises(RuntimeError):
        f(torch.randn(1, 2, 4, 8, 8), torch.randn(1, 3, 5), torch.eye(3).view(1, 1, 3, 3).expand(1, 2, 3, 3), None)
    with pytest.raises(NotImplementedError):
        FeatureFetcher(mode="nearest")
    with pytest.raises(RuntimeError):   # more views than the fused kernel is built for
        fetch_variance(torch.randn(1, 9, 4, 8, 8, device=dev), torch.randn(1, 3, 5, device=dev),
                       torch.eye(3, device=dev).view(1, 1, 3, 3).expand(1, 9, 3, 3), None)


def test_resize_bilinear_vs_interpolate(dev):
    g = torch.Generator().manual_seed(4)
    for (ih, iw, oh, ow) in [(32, 40, 16, 20), (16, 20, 32, 40), (8, 10, 32, 40), (12, 16, 12, 16), (7, 9, 20, 31)]:
        x = torch.randn(3, 5, ih, iw, generator=g)
        ref = F.interpolate(x, (oh, ow), mode="bilinear", align_corners=False)
        out = pointflow.resize_maps(x.to(dev), oh, ow)
        err = _maxabs(out, ref)
        report("resize_%dx%d_%dx%d" % (ih, iw, oh, ow), err=err)
        assert err < 2e-6 * float(x.abs().max())


def test_flow_pyramid_vs_interpolate(dev):
    """One launch, three levels (down-sample, copy, up-sample), channel-last output == F.interpolate + permute."""
    g = torch.Generator().manual_seed(5)
    for V, (h, w), shapes in [(3, (16, 20), [(16, 32, 40), (32, 16, 20), (64, 8, 10)]),
                              (2, (12, 18), [(8, 24, 36), (4, 7, 9), (12, 12, 18)]),
                              (1, (5, 7), [(4, 5, 7), (0, 3, 3), (8, 2, 3)])]:
        maps = [torch.randn(V, c, ih, iw, generator=g) for c, ih, iw in shapes]
        outs = pointflow.flow_pyramid([m.to(dev) for m in maps], h, w)
        for m, o in zip(maps, outs):
            assert o.shape == (V, h, w, m.shape[1])
            if m.shape[1] == 0:
                continue
            ref = F.interpolate(m, (h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
            err = _maxabs(o, ref)
            report("pyramid_%dx%d_%dx%d" % (m.shape[2], m.shape[3], h, w), err=err)
            assert err < 2e-6 * float(m.abs().max())
            # same arithmetic as the planar resize kernel: bit-identical
            assert torch.equal(o.permute(0, 3, 1, 2), pointflow.resize_maps(m.to(dev), h, w))
    lib = _lib.load()
    z = torch.zeros(64, device=dev)
    assert lib.pf_flow_pyramid_f32(_lib.ptr(z), 6, 2, 2, _lib.ptr(z), 4, 2, 2, _lib.ptr(z), 4, 2, 2, 1, 2, 2,
                                   _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), None, None, _lib.stream()) == -2   # c % 4 != 0


def test_flow_pyramid_applies_a_pending_batchnorm_before_interpolating(dev):
    """RawLevel: the tower's raw convolution output + (scale, shift) rows per view == resizing relu(x * scale + shift);
    a level without rows (the plain last convolution of the tower) passes through."""
    g = torch.Generator().manual_seed(8)
    V, (h, w) = 3, (16, 20)
    raws = [torch.randn(V, c, ih, iw, generator=g) for c, ih, iw in [(16, 32, 40), (32, 16, 20), (64, 8, 10)]]
    rows = [(torch.rand(V, r.shape[1], generator=g) + 0.5, torch.randn(V, r.shape[1], generator=g) * 0.3) for r in raws]
    levels = [pointflow.RawLevel(raws[0].to(dev), (rows[0][0].to(dev), rows[0][1].to(dev))),
              pointflow.RawLevel(raws[1].to(dev), (rows[1][0].to(dev), rows[1][1].to(dev))),
              pointflow.RawLevel(raws[2].to(dev), None)]
    outs = pointflow.flow_pyramid(levels, h, w)
    for i, (r, o) in enumerate(zip(raws, outs)):
        x = r.double()
        if i < 2:
            x = torch.relu(x * rows[i][0].double()[:, :, None, None] + rows[i][1].double()[:, :, None, None])
        ref = F.interpolate(x, (h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        err = _maxabs(o, ref)
        report("pyramid_raw_level%d" % i, err=err)
        assert err < 2e-6 * float(x.abs().max())


# ---------------------------------------------------------------------------------------------
# row S
# ---------------------------------------------------------------------------------------------
def test_softargmin_prob_vs_oracle(dev):
    g = torch.Generator().manual_seed(6)
    B, D, H, W = 2, 48, 16, 20
    cost = torch.randn(B, D, H, W, generator=g) * 2.0
    start = torch.tensor([425.0, 500.0])
    interval = torch.tensor([10.6, 5.325])
    end = start + (D - 1) * interval
    depth, prob_vol = O.soft_argmin(cost, start, end, D)
    pm = O.probability_map(prob_vol, depth, start, interval)
    d2, p2 = pointflow.soft_argmin_prob(cost.to(dev), start.to(dev), end.to(dev), interval.to(dev))
    e_d, e_p = _maxabs(d2, depth), _maxabs(p2, pm)
    report("softargmin", depth_err=e_d, prob_err=e_p)
    assert e_d < 1e-6 * float(depth.abs().max()) * 4            # depth within ~4 ulp (1e-4 rel is the contract)
    frac = ((depth - start.view(-1, 1, 1, 1)) / interval.view(-1, 1, 1, 1))
    safe = ((frac - frac.round()).abs() > 1e-3)                 # floor/ceil are discontinuous at integers
    assert float((p2.cpu() - pm).abs()[safe].max()) < 5e-6


# ---------------------------------------------------------------------------------------------
# rows E0 / E1 / E2 and the GEMM building block
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("point_major", [False, True])
@pytest.mark.parametrize("G,Ng,K,cout", [(1, 64, 32, 64), (2, 100, 136, 64), (3, 1000, 224, 64), (1, 333, 64, 128),
                                         (2, 65, 64, 16), (1, 70, 7, 32)])
def test_pointwise_gemm_vs_fp64(dev, point_major, G, Ng, K, cout):
    gen = torch.Generator().manual_seed(G * Ng + K)
    w = torch.randn(cout, K, 1, generator=gen)
    if point_major:
        ldx = K + 4
        x = torch.randn(G * Ng, ldx, generator=gen)
        xm = x[:, :K].view(G, Ng, K)
    else:
        ldx = 0
        x = torch.randn(G, K, Ng, generator=gen)
        xm = x.transpose(1, 2)
    sc = torch.rand(G, K, generator=gen) + 0.5
    sh = torch.randn(G, K, generator=gen) * 0.3
    for affine in (None, (sc, sh)):
        a = xm.double()
        if affine is not None:
            a = torch.relu(a * sc.double().unsqueeze(1) + sh.double().unsqueeze(1))
        ref = a @ w[:, :, 0].double().t()                                    # (G,Ng,cout)
        Wt, _ = pointflow.pack_weight_t(w.to(dev))
        Y = torch.full((G * Ng, cout + 3), -7.0, device=dev)
        aff = None if affine is None else (sc.to(dev), sh.to(dev))
        part = pointflow.pointwise_gemm(x.to(dev), point_major, ldx, Wt, Y, cout + 3, G, Ng, K, cout,
                                        in_affine=aff, want_stats=True)
        out = Y[:, :cout].view(G, Ng, cout)
        scale = float(ref.abs().max())
        err = _maxabs(out, ref)
        report("gemm_K%d_c%d_pm%d_aff%d" % (K, cout, int(point_major), int(affine is not None)), err=err, scale=scale)
        assert err < 2e-6 * scale * max(1.0, (K / 32.0) ** 0.5)              # float32 fma chain of length K
        assert float((Y[:, cout:] + 7.0).abs().max()) == 0.0                 # padding columns untouched
        sums = part.sum(dim=1).cpu()                                         # (G, Nc, 2)
        assert torch.allclose(sums[:, :cout, 0], ref.sum(dim=1), rtol=1e-5, atol=1e-4 * scale)
        assert torch.allclose(sums[:, :cout, 1], (ref ** 2).sum(dim=1), rtol=1e-5)


@pytest.mark.parametrize("name,concat", [("edgeconv_noc", False), ("edgeconv_32", True), ("edgeconv_64", True)])
def test_edgeconv_fused_vs_reference(dev, name, concat):
    g = load_golden(name)
    cin = g["x"].shape[1]
    cout = g["y"].shape[1] // (2 if concat else 1)
    mod = (EdgeConv if concat else EdgeConvNoC)(cin, cout)
    synthetic.seed_weights(mod, seed=1)
    mod = mod.to(dev).train()
    with torch.no_grad():
        y = mod(g["x"].to(dev), g["idx"].to(dev))
    assert y.shape == g["y"].shape and y.is_contiguous()
    err = _maxabs(y, g["y"])
    e_rm = _maxabs(mod.bn.running_mean, g["running_mean"])
    e_rv = float(((mod.bn.running_var.cpu() - g["running_var"]).abs() / g["running_var"]).max())
    report("edgeconv_fused_" + name, err=err, rm_err=e_rm, rv_rel_err=e_rv, scale=float(g["y"].abs().max()))
    # BN batch statistics are reduced in a different order (float64 partials): 2e-5 abs on O(1) activations
    assert err < 2e-5 * max(1.0, float(g["y"].abs().max()))
    assert e_rm < 1e-5 and e_rv < 1e-5
    assert int(mod.bn.num_batches_tracked) == int(g["num_batches_tracked"])
    # run-to-run bit reproducibility (no float atomics in the fused path)
    mod2 = (EdgeConv if concat else EdgeConvNoC)(cin, cout)
    synthetic.seed_weights(mod2, seed=1)
    mod2 = mod2.to(dev).train()
    with torch.no_grad():
        assert torch.equal(mod2(g["x"].to(dev), g["idx"].to(dev)), y)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name,concat", [("edgeconv_noc", False), ("edgeconv_32", True), ("edgeconv_64", True)])
def test_edgeconv_autograd_path_vs_oracle(dev, name, concat, fused, monkeypatch):
    """Training: gradients w.r.t. the input, both 1x1 convs and the BatchNorm affine parameters against the CPU
    oracle's autograd (the reference's composition, networks.py:18-45) -- for the fused autograd node (recompute
    backward kernels, csrc/edgeconv.hip) and for the composed path on the HIP gather_knn operator."""
    from pointmvsnet_amd import networks
    monkeypatch.setattr(networks, "FUSED_TRAIN", 1 if fused else 0)
    g = load_golden(name)
    cin = g["x"].shape[1]
    cout = g["y"].shape[1] // (2 if concat else 1)
    mod = (EdgeConv if concat else EdgeConvNoC)(cin, cout)
    synthetic.seed_weights(mod, seed=1)
    # oracle gradients on the CPU
    sd = {"m." + k: v.clone().requires_grad_(v.is_floating_point() and v.dim() > 0 and "running" not in k)
          for k, v in mod.state_dict().items()}
    x_ref = g["x"].clone().requires_grad_(True)
    y_ref = O.edge_conv(x_ref, g["idx"], sd, "m", concat)
    go = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(9))
    y_ref.backward(go)
    mod = mod.to(dev).train()
    x = g["x"].to(dev).requires_grad_(True)
    y = mod(x, g["idx"].to(dev))
    assert _maxabs(y, g["y"]) < 2e-5 * max(1.0, float(g["y"].abs().max()))
    y.backward(go.to(dev))
    errs = {"dx": _maxabs(x.grad, x_ref.grad) / float(x_ref.grad.abs().max())}
    for pname, p in (("conv1.weight", mod.conv1.weight), ("conv2.weight", mod.conv2.weight),
                     ("bn.weight", mod.bn.weight), ("bn.bias", mod.bn.bias)):
        ref = sd["m." + pname].grad
        assert p.grad is not None and p.grad.shape == ref.shape, pname
        errs[pname.replace(".", "_")] = _maxabs(p.grad, ref) / float(ref.abs().max())
    report("edgeconv_autograd_%s_fused%d" % (name, int(fused)), **errs)
    assert max(errs.values()) < 2e-4, errs
    assert int(mod.bn.num_batches_tracked) == 1
    assert _maxabs(mod.bn.running_mean, g["running_mean"]) < 1e-5


@pytest.mark.parametrize("G,Ng,k", [(1, 700, 16), (3, 257, 16), (2, 90, 5)])
def test_knn_inverse_lists_equal_a_stable_argsort(dev, G, Ng, k):
    """pf_knn_inverse: pair ids stably sorted by target row (with the forward's clamp of out-of-range indices) and the
    list starts, against NumPy's stable argsort -- integers, bit-exact; untargeted rows get empty lists."""
    gen = torch.Generator().manual_seed(G * Ng + k)
    idx = torch.randint(0, Ng // 2, (G, Ng, k), generator=gen)          # half of the points are nobody's neighbour
    idx[:, ::7, 0] = -3                                                # the forward clamps these to 0 / Ng - 1
    idx[:, ::11, 1] = Ng + 5
    order, start = pointflow.knn_inverse(idx.to(dev).contiguous(), G, Ng, k)
    torch.cuda.synchronize()
    keys = (idx.clamp(0, Ng - 1) + torch.arange(G).view(G, 1, 1) * Ng).reshape(-1).numpy()
    want_order = np.argsort(keys, kind="stable")
    want_start = np.searchsorted(keys[want_order], np.arange(G * Ng + 1), side="left")
    assert np.array_equal(order.cpu().numpy().astype(np.int64), want_order)
    assert np.array_equal(start.cpu().numpy().astype(np.int64), want_start)
    # the same tensor again: served from the cache (same objects)
    again = pointflow.knn_inverse(idx.to(dev).contiguous(), G, Ng, k)
    assert again[0].shape == order.shape


@pytest.mark.parametrize("Ng,k,hub", [(3000, 16, "all"), (5000, 16, "most"), (700, 3, "all")])
def test_knn_inverse_hub_rows_are_sorted_in_bounded_work(dev, Ng, k, hub):
    """Every pair (or 90 % of them, the rest random) names ONE row -- all-equal indices, or out-of-range indices
    clamped to a border row: a list of 2 100 .. 48 000 ids, far beyond the LDS rank sort's 1 024.  The chunk + merge
    path must give the stable argsort exactly (round 3's one-thread insertion sort was quadratic here: ADVICE r3)."""
    gen = torch.Generator().manual_seed(Ng + k)
    idx = torch.full((1, Ng, k), Ng + 9, dtype=torch.int64)             # clamped to Ng - 1 by the forward
    if hub == "most":
        mask = torch.rand(1, Ng, k, generator=gen) < 0.1
        idx[mask] = torch.randint(0, Ng, (int(mask.sum()),), generator=gen)
    order, start = pointflow.knn_inverse(idx.to(dev).contiguous(), 1, Ng, k)
    torch.cuda.synchronize()
    keys = idx.clamp(0, Ng - 1).reshape(-1).numpy()
    want_order = np.argsort(keys, kind="stable")
    want_start = np.searchsorted(keys[want_order], np.arange(Ng + 1), side="left")
    assert np.array_equal(start.cpu().numpy().astype(np.int64), want_start)
    assert np.array_equal(order.cpu().numpy().astype(np.int64), want_order)


@pytest.mark.parametrize("cls,cin,cout", [(EdgeConvNoC, 40, 32), (EdgeConv, 32, 32), (EdgeConv, 64, 64)])
def test_edgeconv_backward_is_bit_reproducible_and_matches_the_scatter(dev, cls, cin, cout, monkeypatch):
    """The fused node's backward with the de rows gathered over the inverted index lists (default): two runs give
    identical bits for every gradient; the float-atomic scatter (the reference's scheme) agrees to rounding."""
    D, H, W = 5, 24, 31
    N = D * H * W
    gen = torch.Generator().manual_seed(cout + cin)
    xyz = torch.randn(1, 3, D, H, W, generator=gen).to(dev)
    idx = get_knn_3d(xyz, 5, knn=16)
    mod = cls(cin, cout)
    synthetic.seed_weights(mod, seed=4)
    mod = mod.to(dev).train()
    x0 = torch.randn(1, cin, N, generator=gen).to(dev)
    go = torch.randn(1, (2 if mod.concat else 1) * cout, N, generator=gen).to(dev)

    def grads():
        mod.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        mod(x, idx).backward(go)
        torch.cuda.synchronize()
        return [x.grad.clone()] + [p.grad.clone() for p in mod.parameters()]

    first, second = grads(), grads()
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    # round 6: the default takes TWO walks (the reduce pass leaves every point's sums, the inverted-list gather finishes
    # dl); the three-walk form (PF_EDGE_BWD_SUMS=0) differs in float32 rounding only, and is bit-reproducible as well
    assert pointflow.EDGE_BWD_SUMS
    monkeypatch.setattr(pointflow, "EDGE_BWD_SUMS", False)
    three, three_again = grads(), grads()
    for a, b, c in zip(first, three, three_again):
        assert torch.equal(b, c)
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    monkeypatch.setattr(pointflow, "DETERMINISTIC_BACKWARD", False)
    scattered = grads()
    for a, b in zip(first, scattered):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    assert _lib.status() == 0


def test_edgeconv_eval_mode_uses_running_stats(dev):
    g = load_golden("edgeconv_32")
    mod = EdgeConv(32, 32)
    synthetic.seed_weights(mod, seed=1)
    sd = {"m." + k: v for k, v in mod.state_dict().items()}
    x, idx = g["x"], g["idx"]
    local = F.conv1d(x, sd["m.conv1.weight"])
    edge = F.conv1d(x, sd["m.conv2.weight"])
    nb = O.gather_knn(edge, idx)
    cen = local.unsqueeze(-1).expand(-1, -1, -1, 16)
    e = torch.cat([cen, nb - cen], 1)
    e = F.batch_norm(e, sd["m.bn.running_mean"], sd["m.bn.running_var"], sd["m.bn.weight"], sd["m.bn.bias"], False)
    ref = F.relu(e).mean(3)
    mod = mod.to(dev).eval()
    with torch.no_grad():
        y = mod(x.to(dev), idx.to(dev))
    assert _maxabs(y, ref) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_volume_conv_vs_reference(dev):
    g = load_golden("volume_conv")
    mod = VolumeConv(64, 8)
    synthetic.seed_weights(mod, seed=2)
    mod = mod.to(dev).train()
    with torch.no_grad():
        y = mod(g["x"].to(dev))
    err = _maxabs(y, g["y"])
    report("volume_conv", err=err, scale=float(g["y"].abs().max()))
    assert err < 2e-4 * float(g["y"].abs().max())       # library conv3d (MIOpen) vs mkldnn accumulation order


# ---------------------------------------------------------------------------------------------
# BatchNorm kernels for the conv stacks, batched-view ImageConv, fused VolumeConv
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,sps", [((3, 8, 64, 80), 1), ((6, 16, 7, 9), 2), ((1, 8, 6, 16, 20), 1),
                                       ((2, 5, 3, 5, 7), 2), ((4, 3, 33), 1), ((3, 4, 160, 200), 1),
                                       ((2, 3, 150, 250), 2)])       # the last two exceed the one-launch budget
@pytest.mark.parametrize("relu", [True, False])
def test_batch_norm_act_vs_torch_per_group(dev, shape, sps, relu):
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=gen) * 2.0 + 0.7
    C = shape[1]
    bn_cls = {3: torch.nn.BatchNorm1d, 4: torch.nn.BatchNorm2d, 5: torch.nn.BatchNorm3d}[len(shape)]
    ref_bn, bn = bn_cls(C), bn_cls(C)
    synthetic.seed_weights(ref_bn, 3)
    synthetic.seed_weights(bn, 3)
    ref_bn.train()
    outs = []
    for g0 in range(0, shape[0], sps):                  # the reference: one module call per group, in order
        y = ref_bn(x[g0:g0 + sps])
        outs.append(F.relu(y) if relu else y)
    ref = torch.cat(outs)
    bn = bn.to(dev).train()
    y = pointflow.batch_norm_act_(x.to(dev).contiguous(), bn, relu, sps)
    pointflow.flush_counters()
    err = _maxabs(y, ref)
    report("bn_act_%s" % "x".join(map(str, shape)), err=err)
    assert err < 5e-6 * max(1.0, float(ref.abs().max()))
    assert _maxabs(bn.running_mean, ref_bn.running_mean) < 1e-6
    assert float(((bn.running_var.cpu() - ref_bn.running_var).abs() / ref_bn.running_var).max()) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == shape[0] // sps
    # decoder skip add in the same pass (VolumeConv's last add): addend + act(bn(x))
    bn3 = bn_cls(C)
    synthetic.seed_weights(bn3, 3)
    bn3 = bn3.to(dev).train()
    addend = torch.randn(shape, generator=gen)
    y3 = pointflow.batch_norm_act_(x.to(dev).contiguous(), bn3, relu, sps, addend=addend.to(dev))
    pointflow.flush_counters()
    assert _maxabs(y3, ref + addend) < 5e-6 * max(1.0, float((ref + addend).abs().max()))
    # affine-rows form (the consumer normalises itself): same statistics, no pass over y
    bn2 = bn_cls(C)
    synthetic.seed_weights(bn2, 3)
    bn2 = bn2.to(dev).train()
    xd = x.to(dev).contiguous()
    sc, sh = pointflow.bn_affine_rows(xd, bn2, sps)
    pointflow.flush_counters()
    G = shape[0] // sps
    yy = xd.view(G, sps, C, -1) * sc.view(G, 1, C, 1) + sh.view(G, 1, C, 1)
    yy = (F.relu(yy) if relu else yy).view(shape)
    assert _maxabs(yy, ref) < 5e-6 * max(1.0, float(ref.abs().max()))
    assert _maxabs(bn2.running_mean, ref_bn.running_mean) < 1e-6


def test_image_conv_batched_views_equals_per_view_calls(dev):
    from pointmvsnet_amd.networks import ImageConv
    gen = torch.Generator().manual_seed(12)
    imgs = torch.randn(2, 3, 3, 64, 96, generator=gen)
    a, b = ImageConv(8), ImageConv(8)
    synthetic.seed_weights(a, 5)
    synthetic.seed_weights(b, 5)
    a, b = a.to(dev).train(), b.to(dev).train()
    with torch.no_grad():
        per_view = [a(imgs[:, v].to(dev)) for v in range(3)]               # reference call pattern (model.py:71-77)
        fused = b.forward_views(imgs.to(dev))
        pointflow.flush_counters()
    assert "conv0" not in fused                        # never consumed by the model: not materialised
    for name in ("conv1", "conv2", "conv3"):
        want = torch.stack([pv[name] for pv in per_view], dim=1)
        err = _maxabs(fused[name], want)
        report("image_conv_views_" + name, err=err, scale=float(want.abs().max()))
        assert fused[name].shape == want.shape
        assert err < 2e-4 * float(want.abs().max())
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.allclose(va.float(), vb.float(), rtol=1e-4, atol=1e-5), ka


def test_volume_conv_fused_vs_reference(dev):
    g = load_golden("volume_conv")
    mod = VolumeConv(64, 8)
    synthetic.seed_weights(mod, seed=2)
    mod = mod.to(dev).train()
    with torch.no_grad():
        y = mod.forward_fused(g["x"].to(dev))
    err = _maxabs(y, g["y"])
    report("volume_conv_fused", err=err, scale=float(g["y"].abs().max()))
    assert err < 2e-4 * float(g["y"].abs().max())


# ---------------------------------------------------------------------------------------------
# row R: MFMA conv3d with fused BN statistics
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,Cin,Cout,D,H,W,stride", [(1, 64, 8, 8, 16, 24, 1), (1, 64, 16, 8, 16, 24, 2),
                                                     (2, 16, 32, 5, 7, 19, 1), (1, 32, 16, 6, 9, 33, 2),
                                                     (1, 4, 3, 3, 3, 3, 1), (1, 16, 16, 1, 1, 1, 1),
                                                     (1, 8, 20, 4, 6, 17, 1), (1, 64, 8, 48, 64, 80, 1),
                                                     (1, 64, 16, 48, 64, 80, 2), (1, 16, 16, 24, 32, 40, 1),
                                                     (1, 8, 8, 96, 120, 160, 2), (2, 8, 5, 3, 7, 18, 1),
                                                     (1, 12, 8, 5, 9, 33, 1), (1, 4, 1, 2, 1, 1, 1)])
def test_conv3d_k3_vs_fp64(dev, N, Cin, Cout, D, H, W, stride):
    gen = torch.Generator().manual_seed(N * 1000 + Cin + Cout + D * H * W)
    x = torch.randn(N, Cin, D, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (27 * Cin) ** 0.5
    ref = F.conv3d(x.double(), w.double(), None, stride, 1)
    y, part = pointflow.conv3d_k3(x.to(dev), w.to(dev), stride, True)
    assert y.shape == ref.shape
    scale = float(ref.abs().max())
    err = _maxabs(y, ref)
    report("conv3d_%d_%d_s%d" % (Cin, Cout, stride), err=err, scale=scale)
    assert err < 3e-6 * scale * max(1.0, (27 * Cin / 256.0) ** 0.5)      # float32 fmaf chain of length 27*Cin
    sums = part.sum(dim=1).cpu()                                         # (N, Cout, 2)
    assert torch.allclose(sums[..., 0], ref.sum(dim=(2, 3, 4)), rtol=1e-5, atol=1e-4 * scale)
    assert torch.allclose(sums[..., 1], (ref ** 2).sum(dim=(2, 3, 4)), rtol=1e-5)
    y2, none = pointflow.conv3d_k3(x.to(dev), w.to(dev), stride, False)
    assert none is None and torch.equal(y2, y)                           # deterministic


@pytest.mark.parametrize("N,Cin,Cout,D,H,W", [(1, 8, 1, 8, 16, 24), (2, 5, 3, 3, 5, 7), (1, 8, 1, 48, 64, 80)])
def test_conv3d_k3_few_vs_fp64(dev, N, Cin, Cout, D, H, W):
    gen = torch.Generator().manual_seed(Cin * 100 + D * H * W)
    x = torch.randn(N, Cin, D, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (27 * Cin) ** 0.5
    ref = F.conv3d(x.double(), w.double(), None, 1, 1)
    y = pointflow.conv3d_k3_few(x.to(dev), w.to(dev))
    err = _maxabs(y, ref)
    report("conv3d_few_%d_%d" % (Cin, Cout), err=err, scale=float(ref.abs().max()))
    assert err < 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,form", [(1, 16, 8, 24, 32, 40, ""), (2, 32, 16, 12, 16, 20, ""), (1, 5, 3, 3, 5, 7, ""),
                                                   (1, 4, 6, 1, 1, 1, ""), (1, 64, 32, 6, 8, 10, ""),
                                                   # the matrix-core form: config 4's data gradient of conv1_0 takes it by
                                                   # the shape rule; forced on ragged shapes (half-empty channel group,
                                                   # cells that are no multiple of 16, two samples) and forced off
                                                   (1, 16, 64, 24, 32, 40, ""), (1, 16, 64, 24, 32, 40, "VALU"),
                                                   (2, 32, 16, 12, 16, 20, "MFMA"), (1, 8, 24, 3, 5, 7, "MFMA"),
                                                   (1, 4, 6, 1, 1, 1, "MFMA"), (1, 16, 8, 24, 32, 40, "MFMA")])
@pytest.mark.parametrize("skip", [False, True])
def test_deconv3d_k3s2_vs_fp64(dev, N, Cin, Cout, D, H, W, form, skip, monkeypatch):
    # VolumeConv decoder rows (reference networks.py:141-143): float64 ConvTranspose3d of (xa + xb)
    if form:
        monkeypatch.setenv("PF_DECONV_" + form, "1")
    gen = torch.Generator().manual_seed(Cin * 100 + D * H * W + int(skip))
    xa = torch.randn(N, Cin, D, H, W, generator=gen)
    xb = torch.randn(N, Cin, D, H, W, generator=gen) if skip else None
    w = torch.randn(Cin, Cout, 3, 3, 3, generator=gen) / (4 * Cin) ** 0.5
    xin = xa + xb if skip else xa
    ref = F.conv_transpose3d(xin.double(), w.double(), None, stride=2, padding=1, output_padding=1)
    y, part = pointflow.deconv3d_k3s2(xa.to(dev), None if xb is None else xb.to(dev), w.to(dev), True)
    assert y.shape == ref.shape == (N, Cout, 2 * D, 2 * H, 2 * W)
    scale = float(ref.abs().max())
    err = _maxabs(y, ref)
    report("deconv3d_%d_%d_%d" % (Cin, Cout, int(skip)), err=err, scale=scale)
    # float32 fmaf chain over <= 8*Cin products vs float64: a few ulp of the largest output
    assert err < 3e-6 * scale
    sums = part.sum(dim=1).cpu()
    assert torch.allclose(sums[..., 0], ref.sum(dim=(2, 3, 4)), rtol=1e-4, atol=1e-3 * scale)
    assert torch.allclose(sums[..., 1], (ref ** 2).sum(dim=(2, 3, 4)), rtol=1e-5)
    y2, none = pointflow.deconv3d_k3s2(xa.to(dev), None if xb is None else xb.to(dev), w.to(dev), False)
    assert none is None and torch.equal(y2, y)                           # deterministic


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks,stride", [(3, 64, 64, 64, 80, 3, 1), (2, 64, 64, 19, 25, 3, 1),
                                                       (3, 32, 64, 128, 160, 5, 2), (1, 32, 64, 37, 51, 5, 2),
                                                       (3, 32, 32, 32, 48, 3, 1), (1, 32, 32, 13, 30, 3, 1),
                                                       (2, 16, 32, 64, 80, 5, 2), (1, 16, 32, 27, 33, 5, 2),
                                                       (1, 64, 64, 3, 5, 3, 1), (3, 16, 16, 64, 96, 3, 1),
                                                       (2, 16, 16, 33, 47, 3, 1), (2, 8, 16, 128, 160, 5, 2),
                                                       (1, 8, 16, 61, 75, 5, 2), (1, 16, 16, 5, 3, 3, 1),
                                                       (3, 3, 8, 64, 96, 3, 1), (1, 3, 8, 37, 51, 3, 1),
                                                       (2, 8, 8, 64, 80, 3, 1), (1, 8, 8, 19, 21, 3, 1)])
@pytest.mark.parametrize("affine", [False, True])
def test_conv2d_wide_vs_fp64(dev, N, Cin, Cout, H, W, ks, stride, affine):
    """csrc/conv2d_wide.hip (the 32x32x2-MFMA mapping for the towers' small maps) against a float64 convolution:
    aligned maps, ragged edges in both directions (masked stores, masked statistics), maps smaller than a tile."""
    gen = torch.Generator().manual_seed(N * 1000 + Cin + Cout + H * W)
    x = torch.randn(N, Cin, H, W, generator=gen)
    conv = torch.nn.Conv2d(Cin, Cout, ks, stride=stride, padding=ks // 2, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) / (ks * ks * Cin) ** 0.5)
    sc = torch.rand(N, Cin, generator=gen) + 0.5
    sh = torch.randn(N, Cin, generator=gen) * 0.3
    xin = x.double()
    if affine:
        xin = torch.relu(xin * sc.double().view(N, Cin, 1, 1) + sh.double().view(N, Cin, 1, 1))
    ref = F.conv2d(xin, conv.weight.double(), None, stride, ks // 2)
    conv = conv.to(dev)
    assert pointflow.conv2d_wide_supported(conv)
    if affine and Cin == 3:
        pytest.skip("the image layer has no pending BatchNorm (PF_ERR_UNSUPPORTED by contract)")
    aff = (sc.to(dev), sh.to(dev)) if affine else None
    y, part = pointflow.conv2d_wide(x.to(dev), conv, aff, 1, True)
    assert y.shape == ref.shape
    scale = float(ref.abs().max())
    err = _maxabs(y, ref)
    report("conv2d_wide_%d_%d_k%d_%dx%d_aff%d" % (Cin, Cout, ks, H, W, int(affine)), err=err, scale=scale)
    assert err < 3e-6 * scale * max(1.0, (ks * ks * Cin / 256.0) ** 0.5)
    sums = part.sum(dim=1).cpu()
    assert torch.allclose(sums[..., 0], ref.sum(dim=(2, 3)), rtol=1e-5, atol=1e-4 * scale)
    assert torch.allclose(sums[..., 1], (ref ** 2).sum(dim=(2, 3)), rtol=1e-5)
    y2, none = pointflow.conv2d_wide(x.to(dev), conv, aff, 1, False)
    assert none is None and torch.equal(y2, y)                           # deterministic
    if Cout >= 32:                                                       # channel-last output: the same numbers
        ycl, pcl = pointflow.conv2d_wide(x.to(dev), conv, aff, 1, True, channel_last_out=True)
        assert ycl.shape == (N, y.shape[2], y.shape[3], Cout)
        assert torch.equal(ycl.permute(0, 3, 1, 2), y) and torch.equal(pcl, part)
    assert _lib.status() == 0


def test_conv2d_wide_per_view_affine_rows(dev):
    """samples_per_stat > 1: sample n takes affine row n // sps (the views of a scene batched along N)."""
    gen = torch.Generator().manual_seed(5)
    N, sps, C = 4, 2, 64
    x = torch.randn(N, C, 12, 20, generator=gen)
    conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False)
    sc = torch.rand(N // sps, C, generator=gen) + 0.5
    sh = torch.randn(N // sps, C, generator=gen) * 0.3
    rows = torch.arange(N) // sps
    xin = torch.relu(x.double() * sc.double()[rows].view(N, C, 1, 1) + sh.double()[rows].view(N, C, 1, 1))
    ref = F.conv2d(xin, conv.weight.double(), None, 1, 1)
    y, _ = pointflow.conv2d_wide(x.to(dev), conv.to(dev), (sc.to(dev), sh.to(dev)), sps, False)
    assert _maxabs(y, ref) < 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("concat,k", [(True, 5), (False, 8), (True, 16)])
def test_edgeconv_fused_arbitrary_indices_vs_first_principles(dev, concat, k):
    """The module API takes ANY (B,N,k) int64 indices (not only lattice windows) and any k."""
    gen = torch.Generator().manual_seed(k)
    B, cin, cout, N = 2, 24, 32, 301
    x = torch.randn(B, cin, N, generator=gen)
    idx = torch.randint(0, N, (B, N, k), generator=gen)
    mod = (EdgeConv if concat else EdgeConvNoC)(cin, cout)
    synthetic.seed_weights(mod, seed=4)
    ref, _ = BF.edge_conv(x.numpy(), idx.numpy(), mod.conv1.weight[:, :, 0].detach().numpy(),
                          mod.conv2.weight[:, :, 0].detach().numpy(), mod.bn.weight.detach().numpy(),
                          mod.bn.bias.detach().numpy(), concat)
    mod = mod.to(dev).train()
    with torch.no_grad():
        y = mod(x.to(dev), idx.to(dev))
    err = float(np.abs(y.cpu().numpy() - ref).max())
    report("edgeconv_arbitrary_k%d" % k, err=err)
    assert err < 3e-5
    assert _lib.status() == 0


@pytest.mark.parametrize("C,G,gps,T", [(10, 6, 2, 7), (64, 4, 1, 200), (3, 1, 1, 1), (257, 5, 5, 9)])
def test_bn_finalize_jobs_vs_first_principles(dev, C, G, gps, T):
    torch.manual_seed(C + G)
    S = G // gps
    jobs, expect = [], []
    for j in range(2):
        pcols = C + 3 * j
        col0 = 2 * j
        part = torch.rand((G, T, pcols, 2), dtype=torch.float64) + 0.5
        part[..., 1] += 40.0                                  # keep the variance positive
        count = 37.0 + j
        bn = torch.nn.BatchNorm1d(C).to(dev)
        with torch.no_grad():
            bn.weight.uniform_(-1.0, 1.0)
            bn.bias.uniform_(-1.0, 1.0)
            bn.running_mean.uniform_(-1.0, 1.0)
            bn.running_var.uniform_(0.5, 2.0)
        rm, rv = bn.running_mean.double().cpu().clone(), bn.running_var.double().cpu().clone()
        scale = torch.full((S, C + 4), 7.0, device=dev)
        shift = torch.full((S, C + 4), 7.0, device=dev)
        pd = part.to(dev)
        jobs.append((pointflow.bn_job(bn, pd, col0, C, count, count, G, gps, scale, shift), pd, bn, scale, shift))
        sums = part[:, :, col0:col0 + C].reshape(S, gps * T, C, 2).sum(1)
        mean = sums[..., 0] / count
        var = (sums[..., 1] / count - mean * mean).clamp_min(0.0)
        a = bn.weight.double().cpu() / torch.sqrt(var + bn.eps)
        b = bn.bias.double().cpu() - mean * a
        for s in range(S):                                    # one running-stat update per stat group, in order
            rm = (1 - bn.momentum) * rm + bn.momentum * mean[s]
            rv = (1 - bn.momentum) * rv + bn.momentum * var[s] * (count / (count - 1.0))
        expect.append((a, b, rm, rv))
    pointflow.bn_finalize_jobs([j[0] for j in jobs])
    torch.cuda.synchronize()
    for (_, _, bn, scale, shift), (a, b, rm, rv) in zip(jobs, expect):
        # float32 outputs of a float64 reduction: tolerance = float32 rounding of the result
        assert torch.allclose(scale[:, :C].double().cpu(), a, rtol=1e-6, atol=1e-7)
        assert torch.allclose(shift[:, :C].double().cpu(), b, rtol=1e-6, atol=1e-6)
        assert torch.all(scale[:, C:] == 7.0) and torch.all(shift[:, C:] == 7.0)   # ld_affine padding untouched
        assert torch.allclose(bn.running_mean.double().cpu(), rm, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_var.double().cpu(), rv, rtol=1e-5, atol=1e-6)
    assert _lib.status() == 0


def test_image_conv_tower_vs_oracle(dev):
    """The batched-views tower (own conv kernels, BatchNorm statistics in the epilogues) against the CPU oracle's
    tower (reference networks.py:84-124 on ATen), per view, including the running statistics."""
    from oracle import pointflow_oracle as O
    from pointmvsnet_amd.networks import ImageConv
    gen = torch.Generator().manual_seed(21)
    imgs = torch.randn(1, 3, 3, 128, 160, generator=gen)
    net = ImageConv(8)
    synthetic.seed_weights(net, 3)
    sd = {"t." + k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev).train()
    with torch.no_grad():
        got = net.forward_views(imgs.to(dev), need=("conv1", "conv2", "conv3"))
        pointflow.flush_counters()
        track = {}
        want = [O.image_conv(imgs[:, v], sd, "t", track) for v in range(3)]
    for name in ("conv1", "conv2", "conv3"):
        w = torch.stack([o[name] for o in want], dim=1)
        err = _maxabs(got[name], w)
        report("image_conv_tower_vs_oracle_" + name, err=err, scale=float(w.abs().max()))
        assert err < 2e-5 * max(1.0, float(w.abs().max()))
    state = net.state_dict()
    assert len(track) == 3 * 10                        # ten BatchNorms, (mean, var, counter) each
    for key, val in track.items():                     # three sequential updates per module, like the reference
        if key.endswith("num_batches_tracked"):
            assert int(state[key[2:]]) == int(val), key
        else:
            assert torch.allclose(state[key[2:]].cpu(), val, rtol=1e-4, atol=1e-5), key


@pytest.mark.parametrize("hw", [(128, 160), (72, 104)])
def test_tower_pair_launches_equal_the_two_towers_bit_for_bit(dev, hw):
    """pf_conv2d_wide_sets_f32 (both towers in each of the eleven launches: weights, pending BatchNorm and output
    layout per set) against the same towers run one after the other: every output the fused forward consumes, the
    pending affine rows of the pyramid levels, and all running statistics -- bit for bit."""
    import copy
    from pointmvsnet_amd.networks import ImageConv, tower_pair_supported, tower_pair_views
    gen = torch.Generator().manual_seed(5)
    imgs = torch.randn(1, 3, 3, hw[0], hw[1], generator=gen).to(dev)
    coarse, flow = ImageConv(8), ImageConv(8)
    synthetic.seed_weights(coarse, 1)
    synthetic.seed_weights(flow, 2)
    coarse, flow = coarse.to(dev).train(), flow.to(dev).train()
    coarse2, flow2 = copy.deepcopy(coarse), copy.deepcopy(flow)
    assert tower_pair_supported(coarse, flow, imgs)
    with torch.no_grad():
        want_c = coarse.forward_views(imgs, need=("conv3",), channel_last=("conv3",))["conv3_cl"]
        want_f = flow.forward_views(imgs, raw=("conv1", "conv2", "conv3"))
        pointflow.flush_counters()
        got_c, got_f = tower_pair_views(coarse2, flow2, imgs)
        pointflow.flush_counters()
    torch.cuda.synchronize()
    assert got_c.shape == want_c.shape and torch.equal(got_c, want_c)
    for name in ("conv1", "conv2", "conv3"):
        w, g = want_f[name + "_raw"], got_f[name]
        assert torch.equal(g.raw, w.raw), name
        if w.affine is None:
            assert g.affine is None
        else:
            for a, b in zip(pointflow.affine_rows(g.affine), pointflow.affine_rows(w.affine)):
                assert torch.equal(a, b), name
    for ref, new in ((coarse, coarse2), (flow, flow2)):
        for (k, a), (_, b) in zip(ref.state_dict().items(), new.state_dict().items()):
            assert torch.equal(a, b), k
    assert int(coarse2.conv1[0].bn.num_batches_tracked) == 3 and _lib.status() == 0


# ---------------------------------------------------------------------------------------------
# lattice kNN: sorting-network kernel, window codes as the neighbourhood of the EdgeConv passes
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 5, 64, 80), (4, 5, 64, 80), (2, 5, 7, 33), (16, 5, 30, 40)])
def test_knn_network_kernel_equals_insertion_kernels_and_codes_only_call(dev, shape):
    B, D, H, W = shape
    g = torch.Generator().manual_seed(B * H)
    base = torch.stack(torch.meshgrid(torch.arange(W).float(), torch.arange(H).float(), torch.arange(D).float(),
                                      indexing="xy"), 0).permute(0, 3, 1, 2)
    xyz = (base.unsqueeze(0).repeat(B, 1, 1, 1, 1) * 0.07 - 1.0 + 0.02 * torch.randn(B, 3, D, H, W, generator=g)).to(dev)
    xyz[:, :, :, :2, :3] = xyz[:, :, :, 2:4, :3]                  # exact duplicates: ties resolved by the code order
    idx, codes = knn_lattice(xyz, 5, 16, with_codes=True)
    none, codes_only = knn_lattice(xyz, 5, 16, with_codes=True, with_idx=False)
    assert none is None and torch.equal(codes_only, codes)
    # k = 20 (get_knn_3d's default) takes the insertion-list kernels: same ranking rule, so the same first 16
    idx20, codes20 = knn_lattice(xyz, 5, 20, with_codes=True)
    assert torch.equal(idx20[:, :, :16], idx) and torch.equal(codes20[:, :, :16], codes)
    bf_idx, bf_code = BF.knn_window(xyz[0].cpu().numpy(), 5, 16)
    assert np.array_equal(idx[0].cpu().numpy(), bf_idx) and np.array_equal(codes[0].cpu().numpy(), bf_code)


@pytest.mark.parametrize("concat,C", [(False, 32), (True, 32), (True, 64)])
def test_edgeconv_window_codes_equal_int64_indices(dev, concat, C):
    """The EdgeConv passes fed with the 16-byte window codes must reproduce, bit for bit, what they give on the
    int64 indices of the same kNN call (incl. the clamp at the ends of each group)."""
    G, D, H, W, K = 3, 5, 9, 14, 40
    Ng = D * H * W
    gen = torch.Generator().manual_seed(C)
    xyz = torch.randn(G, 3, D, H, W, generator=gen).to(dev)       # random cloud: neighbours all over the window
    idx, codes = knn_lattice(xyz, 5, 16, with_codes=True)
    X = torch.randn(G * Ng, K, generator=gen).to(dev)
    outs = []
    for use_codes in (False, True):
        mod = (EdgeConv if concat else EdgeConvNoC)(K, C)
        synthetic.seed_weights(mod, seed=2)
        mod = mod.to(dev).train()
        width = (2 if concat else 1) * C
        Y = torch.empty((G * Ng, width), device=dev)
        with torch.no_grad():
            pointflow.edge_conv_fused(X, True, K, K, G, Ng, None if use_codes else idx, mod.conv1.weight,
                                      mod.conv2.weight, mod.bn, concat, Y, width,
                                      codes=codes if use_codes else None, lattice=(5, H, W) if use_codes else None)
            pointflow.flush_counters()
        outs.append((Y, mod.bn.running_mean.clone(), mod.bn.running_var.clone()))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert _lib.status() == 0


# ---------------------------------------------------------------------------------------------
# coarse warp on channel-last maps (csrc/fetch.hip, frustum_variance_cl_kernel)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,V,C,H,W,D", [(1, 3, 64, 64, 80, 48), (2, 5, 8, 16, 20, 6), (1, 7, 68, 9, 11, 3),
                                         (1, 2, 4, 5, 7, 1)])
def test_frustum_variance_channel_last_is_bit_identical(dev, B, V, C, H, W, D):
    from pointmvsnet_amd.utils.feature_fetcher import frustum_variance, to_channel_last
    gen = torch.Generator().manual_seed(C + H)
    data = synthetic.make_scene(8 * H, 8 * W, V, D, seed=V, batch=B)
    cams = data["cam_params_list"]
    K = cams[:, :, 1, :3, :3].clone()
    K[:, :, :2, :3] /= 8.0
    E = cams[:, :, 0, :3, :4].clone()
    feats = torch.randn(B, V, C, H, W, generator=gen).to(dev)
    kinv = torch.inverse(K[:, 0]).to(dev)
    rinv = torch.inverse(E[:, 0, :, :3]).to(dev)
    t0 = E[:, 0, :, 3].to(dev)
    depths = torch.stack([torch.linspace(425.0 + 10 * b, 900.0 + 10 * b, D) for b in range(B)]).to(dev)
    cl = to_channel_last(feats)
    assert cl.shape == (B, V, H, W, C) and torch.equal(cl, feats.permute(0, 1, 3, 4, 2).contiguous())
    a, wa = frustum_variance(feats, kinv, rinv, t0, depths, K.to(dev), E.to(dev), channel_last=False)
    b, wb = frustum_variance(feats, kinv, rinv, t0, depths, K.to(dev), E.to(dev), channel_last=True)
    assert torch.equal(a, b) and torch.equal(wa, wb)
    c, none = frustum_variance(feats, kinv, rinv, t0, depths, K.to(dev), E.to(dev), want_points=False,
                               channel_last=True)
    assert none is None and torch.equal(c, a)


# ---------------------------------------------------------------------------------------------
# pointwise GEMM, direct-A kernel (the PointFlow chain's six shapes) against the chunked kernel and fp64
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,cout,ldx", [(136, 64, 136), (32, 64, 224), (64, 128, 224), (224, 64, 224), (64, 64, 64),
                                        (64, 16, 64)])
@pytest.mark.parametrize("G,Ng", [(1, 25600), (4, 1000), (3, 129), (2, 31)])
def test_pointwise_gemm_direct_kernel(dev, K, cout, ldx, G, Ng):
    gen = torch.Generator().manual_seed(K + cout + Ng)
    w = torch.randn(cout, K, 1, generator=gen)
    x = torch.randn(G * Ng, ldx, generator=gen)
    sc = torch.rand(G, K, generator=gen) + 0.5
    sh = torch.randn(G, K, generator=gen) * 0.3
    Wt, _ = pointflow.pack_weight_t(w.to(dev))
    xd = x.to(dev)
    for affine in (None, (sc.to(dev), sh.to(dev))):
        a = x[:, :K].view(G, Ng, K).double()
        if affine is not None:
            a = torch.relu(a * sc.double().unsqueeze(1) + sh.double().unsqueeze(1))
        ref = a @ w[:, :, 0].double().t()
        outs = []
        x_cm = xd[:, :K].reshape(G, Ng, K).transpose(1, 2).contiguous()      # channel-major: the chunked kernel
        for point_major in (False, True):
            Y = torch.full((G * Ng, cout + 4), -7.0, device=dev)
            part = pointflow.pointwise_gemm(xd if point_major else x_cm, point_major, ldx if point_major else 0, Wt,
                                            Y, cout + 4, G, Ng, K, cout, in_affine=affine, want_stats=True)
            torch.cuda.synchronize()
            outs.append((Y, part))
        scale = float(ref.abs().max())
        for Y, part in outs:
            assert _maxabs(Y[:, :cout].view(G, Ng, cout), ref) < 2e-6 * scale * max(1.0, (K / 32.0) ** 0.5)
            assert float((Y[:, cout:] + 7.0).abs().max()) == 0.0
            sums = part.sum(dim=1).cpu()
            assert torch.allclose(sums[:, :cout, 0], ref.sum(dim=1), rtol=1e-5, atol=1e-4 * scale)
            assert torch.allclose(sums[:, :cout, 1], (ref ** 2).sum(dim=1), rtol=1e-5)
        # the two kernels order the K sum differently: equal to float32 rounding
        assert _maxabs(outs[0][0], outs[1][0]) < 4e-6 * scale
    assert _lib.status() == 0


# ---------------------------------------------------------------------------------------------
# BatchNorm finalize folded into the CONSUMER (pf_bn_resolve, csrc/pf_bn_resolve.h): same numbers as the separate
# finalize launch, running statistics from the deferred batched finalize
# ---------------------------------------------------------------------------------------------
def _bn_pair(C, dev):
    a, b = torch.nn.BatchNorm1d(C).to(dev).train(), torch.nn.BatchNorm1d(C).to(dev).train()
    with torch.no_grad():
        for m in (a, b):
            m.weight.copy_(torch.linspace(0.5, 1.5, C))
            m.bias.copy_(torch.linspace(-0.3, 0.3, C))
    return a, b


@pytest.mark.parametrize("G,Ng", [(16, 1600), (4, 6400), (3, 129), (1, 25600)])
@pytest.mark.parametrize("K,cout", [(64, 64), (64, 16)])
def test_gemm_resolves_pending_batchnorm_like_the_finalize_launch(dev, G, Ng, K, cout):
    gen = torch.Generator().manual_seed(G * 7 + Ng + cout)
    w0 = torch.randn(K, 40, 1, generator=gen).to(dev)
    w1 = torch.randn(cout, K, 1, generator=gen).to(dev)
    x = torch.randn(G * Ng, 40, generator=gen).to(dev)
    W0, _ = pointflow.pack_weight_t(w0)
    W1, _ = pointflow.pack_weight_t(w1)
    Z = torch.empty((G * Ng, K), device=dev)
    part = pointflow.pointwise_gemm(x, True, 40, W0, Z, K, G, Ng, 40, K, want_stats=True)
    bn_rows, bn_lazy = _bn_pair(K, dev)
    rows = pointflow._bn_affine_from_gemm(bn_rows, part, K, G, Ng, 1, dev)
    lazy = pointflow._bn_affine_from_gemm(bn_lazy, part, K, G, Ng, 1, dev, lazy=True)
    assert isinstance(lazy, pointflow.LazyAffine) and not isinstance(rows, pointflow.LazyAffine)
    outs = []
    for aff in (rows, lazy):
        Y = torch.empty((G * Ng, cout), device=dev)
        pointflow.pointwise_gemm(Z, True, K, W1, Y, cout, G, Ng, K, cout, in_affine=aff)
        outs.append(Y)
    scale = float(outs[0].abs().max())
    assert _maxabs(outs[0], outs[1]) < 2e-6 * scale          # (scale, shift) agree to float rounding
    # the chunked kernel has no in_bn slot: the call materialises the rows itself (no running-stat update)
    bn_c, _ = _bn_pair(K, dev)
    lazy2 = pointflow._bn_affine_from_gemm(bn_c, part, K, G, Ng, 1, dev, lazy=True)
    Y2 = torch.empty((G * Ng, cout), device=dev)
    Z_cm = Z.view(G, Ng, K).transpose(1, 2).contiguous()                     # channel-major input: chunked kernel
    pointflow.pointwise_gemm(Z_cm, False, 0, W1, Y2, cout, G, Ng, K, cout, in_affine=lazy2)
    assert _maxabs(Y2, outs[0]) < 6e-6 * scale
    # running statistics: untouched until the deferred finalize, then exactly the separate launch's
    assert float(bn_lazy.running_mean.abs().max()) == 0.0
    pointflow.flush_lazy_stats(dev)
    pointflow.flush_counters()
    torch.cuda.synchronize()
    for b in (bn_lazy, bn_c):
        assert torch.equal(b.running_mean, bn_rows.running_mean) and torch.equal(b.running_var, bn_rows.running_var)
        assert int(b.num_batches_tracked) == G
    assert torch.equal(lazy.scale, rows[0]) and torch.equal(lazy.shift, rows[1])
    assert _lib.status() == 0


@pytest.mark.parametrize("N,sps,Cin,Cout,H,W,ks,stride", [(3, 1, 32, 32, 128, 160, 3, 1), (4, 2, 64, 64, 23, 37, 3, 1),
                                                           (3, 1, 32, 64, 64, 80, 5, 2)])
def test_conv2d_wide_resolves_pending_batchnorm(dev, N, sps, Cin, Cout, H, W, ks, stride):
    """producer conv (wide kernel, 3x3 Cin -> Cin, statistics in its epilogue) -> BatchNorm pending -> the consumer
    conv resolves it in its prologue."""
    gen = torch.Generator().manual_seed(N + Cin + H)
    prod = torch.nn.Conv2d(Cin, Cin, 3, padding=1, bias=False).to(dev)
    cons = torch.nn.Conv2d(Cin, Cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
    x = torch.randn(N, Cin, H, W, generator=gen).to(dev)
    y0, part = pointflow.conv2d_wide(x, prod, None, sps, True)
    bn_rows, bn_lazy = torch.nn.BatchNorm2d(Cin).to(dev).train(), torch.nn.BatchNorm2d(Cin).to(dev).train()
    with torch.no_grad():
        for m in (bn_rows, bn_lazy):
            m.weight.copy_(torch.linspace(0.5, 1.5, Cin))
            m.bias.copy_(torch.linspace(-0.3, 0.3, Cin))
    rows = pointflow.bn_affine_rows(y0, bn_rows, sps, part)
    lazy = pointflow.bn_affine_rows(y0, bn_lazy, sps, part, lazy=True)
    assert isinstance(lazy, pointflow.LazyAffine)
    ya, _ = pointflow.conv2d_wide(y0, cons, rows, sps, False)
    yb, pb = pointflow.conv2d_wide(y0, cons, lazy, sps, True)
    assert _maxabs(ya, yb) < 2e-6 * float(ya.abs().max())
    # against torch: BatchNorm per stat group on the producer's output, ReLU, float64 convolution
    G = N // sps
    ref_in = torch.cat([torch.relu(F.batch_norm(y0[g * sps:(g + 1) * sps].double(), None, None, bn_rows.weight.double(),
                                                bn_rows.bias.double(), True, 0.0, bn_rows.eps)) for g in range(G)])
    ref = F.conv2d(ref_in, cons.weight.double(), None, stride, ks // 2)
    assert _maxabs(yb, ref) < 2e-5 * float(ref.abs().max())
    pointflow.flush_lazy_stats(dev)
    pointflow.flush_counters()
    torch.cuda.synchronize()
    assert torch.equal(bn_lazy.running_mean, bn_rows.running_mean) and torch.equal(bn_lazy.running_var, bn_rows.running_var)
    assert _lib.status() == 0


# ---------------------------------------------------------------------------------------------
# the bottom of VolumeConv's U-Net (csrc/conv3d_bottom.hip) against float64 convolutions
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,Cin,stride,D,H,W", [(1, 32, 2, 12, 32, 40), (2, 32, 2, 5, 7, 9), (1, 64, 1, 6, 16, 20),
                                                (2, 64, 1, 3, 5, 7), (1, 64, 1, 6, 8, 10)])
@pytest.mark.parametrize("affine", ["none", "rows", "lazy"])
def test_conv3d_bottom_vs_fp64(dev, N, Cin, stride, D, H, W, affine):
    gen = torch.Generator().manual_seed(N + Cin + D * H * W)
    conv = torch.nn.Conv3d(Cin, 64, 3, stride=stride, padding=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) / (27 * Cin) ** 0.5)
    x = torch.randn(N, Cin, D, H, W, generator=gen)
    assert pointflow.conv3d_bottom_supported(conv)
    xin, aff = x.double(), None
    if affine == "rows":
        sc = torch.rand(1, Cin, generator=gen) + 0.5
        sh = torch.randn(1, Cin, generator=gen) * 0.3
        xin = torch.relu(x.double() * sc.double().view(1, Cin, 1, 1, 1) + sh.double().view(1, Cin, 1, 1, 1))
        aff = (sc.to(dev), sh.to(dev))
    xd = x.to(dev)
    if affine == "lazy":
        # x is the raw output of a producer with statistics rows: here a statistics pass over x itself
        bn = torch.nn.BatchNorm3d(Cin).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, Cin))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, Cin))
        S = D * H * W
        T = 7
        part = torch.zeros((N, T, Cin, 2), dtype=torch.float64, device=dev)
        chunks = torch.chunk(xd.double().reshape(N, Cin, S), T, dim=2)
        for t, ch in enumerate(chunks):
            part[:, t, :, 0] = ch.sum(dim=2)
            part[:, t, :, 1] = (ch * ch).sum(dim=2)
        aff = pointflow.bn_affine_rows(xd, bn, N, part, lazy=True)
        assert isinstance(aff, pointflow.LazyAffine)
        xin = torch.relu(F.batch_norm(x.double(), None, None, bn.weight.double().cpu(), bn.bias.double().cpu(), True, 0.0,
                                      bn.eps))
    ref = F.conv3d(xin, conv.weight.double(), None, stride, 1)
    y, part_y = pointflow.conv3d_bottom(xd, conv.to(dev), aff, N, True)
    assert y.shape == ref.shape
    scale = float(ref.abs().max())
    err = _maxabs(y, ref)
    report("conv3d_bottom_%d_s%d_%dx%dx%d_%s" % (Cin, stride, D, H, W, affine), err=err, scale=scale)
    assert err < (2e-5 if affine == "lazy" else 4e-6) * scale * max(1.0, (27 * Cin / 256.0) ** 0.5)
    sums = part_y.sum(dim=1).cpu()
    assert torch.allclose(sums[..., 0], ref.sum(dim=(2, 3, 4)), rtol=1e-4, atol=2e-4 * scale * ref[0, 0].numel() ** 0.5)
    assert torch.allclose(sums[..., 1], (ref ** 2).sum(dim=(2, 3, 4)), rtol=1e-4)
    y2, none = pointflow.conv3d_bottom(xd, conv, aff if affine != "lazy" else aff.rows(), N, False)
    assert none is None and (affine == "lazy" or torch.equal(y2, y))
    pointflow.flush_counters()
    assert _lib.status() == 0


@pytest.mark.parametrize("N,D,H,W", [(1, 6, 16, 20), (2, 3, 5, 7), (1, 1, 4, 4), (1, 6, 8, 10)])
@pytest.mark.parametrize("affine", [False, True])
def test_deconv3d_bottom_vs_fp64(dev, N, D, H, W, affine):
    gen = torch.Generator().manual_seed(N + D * H * W)
    conv = torch.nn.ConvTranspose3d(64, 32, 3, stride=2, padding=1, output_padding=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) / (27 * 64 / 8) ** 0.5)
    x = torch.randn(N, 64, D, H, W, generator=gen)
    assert pointflow.deconv3d_bottom_supported(conv)
    xin, aff = x.double(), None
    if affine:
        sc = torch.rand(1, 64, generator=gen) + 0.5
        sh = torch.randn(1, 64, generator=gen) * 0.3
        xin = torch.relu(x.double() * sc.double().view(1, 64, 1, 1, 1) + sh.double().view(1, 64, 1, 1, 1))
        aff = (sc.to(dev), sh.to(dev))
    ref = F.conv_transpose3d(xin, conv.weight.double(), None, 2, 1, 1)
    y, part = pointflow.deconv3d_bottom(x.to(dev), conv.to(dev), aff, N, True)
    assert y.shape == ref.shape
    scale = float(ref.abs().max())
    err = _maxabs(y, ref)
    report("deconv3d_bottom_%dx%dx%d_aff%d" % (D, H, W, int(affine)), err=err, scale=scale)
    assert err < 4e-6 * scale
    sums = part.sum(dim=1).cpu()
    assert torch.allclose(sums[..., 0], ref.sum(dim=(2, 3, 4)), rtol=1e-4, atol=2e-4 * scale * ref[0, 0].numel() ** 0.5)
    assert torch.allclose(sums[..., 1], (ref ** 2).sum(dim=(2, 3, 4)), rtol=1e-4)
    y2, none = pointflow.deconv3d_bottom(x.to(dev), conv, aff, N, False)
    assert none is None and torch.equal(y2, y)
    assert _lib.status() == 0


@pytest.mark.parametrize("N,Cin,Cout,stride,D,H,W", [(1, 16, 32, 2, 24, 32, 40), (2, 16, 16, 1, 7, 9, 21), (1, 32, 32, 1, 12, 16, 20)])
@pytest.mark.parametrize("affine", ["rows", "lazy"])
def test_conv3d_k3_applies_pending_batchnorm_while_staging(dev, N, Cin, Cout, stride, D, H, W, affine):
    """pf_conv3d_k3_f32 with the input's BatchNorm + ReLU pending (rows, or resolved by the launch from statistics
    rows) against BatchNorm -> ReLU -> float64 convolution; zero padding applies AFTER the activation."""
    gen = torch.Generator().manual_seed(N + Cin + Cout + D * H * W)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (27 * Cin) ** 0.5
    x = torch.randn(N, Cin, D, H, W, generator=gen)
    xd = x.to(dev)
    if affine == "rows":
        sc = torch.rand(1, Cin, generator=gen) + 0.5
        sh = torch.randn(1, Cin, generator=gen) * 0.3
        xin = torch.relu(x.double() * sc.double().view(1, Cin, 1, 1, 1) + sh.double().view(1, Cin, 1, 1, 1))
        aff = (sc.to(dev), sh.to(dev))
    else:
        bn = torch.nn.BatchNorm3d(Cin).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, Cin))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, Cin))
        T = 5
        part = torch.zeros((N, T, Cin, 2), dtype=torch.float64, device=dev)
        for t, ch in enumerate(torch.chunk(xd.double().reshape(N, Cin, -1), T, dim=2)):
            part[:, t, :, 0] = ch.sum(dim=2)
            part[:, t, :, 1] = (ch * ch).sum(dim=2)
        aff = pointflow.bn_affine_rows(xd, bn, N, part, lazy=True)
        assert isinstance(aff, pointflow.LazyAffine)
        xin = torch.relu(F.batch_norm(x.double(), None, None, bn.weight.double().cpu(), bn.bias.double().cpu(), True, 0.0,
                                      bn.eps))
    ref = F.conv3d(xin, w.double(), None, stride, 1)
    y, part_y = pointflow.conv3d_k3(xd, w.to(dev), stride, True, in_affine=aff, samples_per_stat=N)
    scale = float(ref.abs().max())
    err = _maxabs(y, ref)
    report("conv3d_k3_affine_%d_%d_s%d_%s" % (Cin, Cout, stride, affine), err=err, scale=scale)
    assert err < (2e-5 if affine == "lazy" else 4e-6) * scale * max(1.0, (27 * Cin / 256.0) ** 0.5)
    sums = part_y.sum(dim=1).cpu()
    assert torch.allclose(sums[..., 0], ref.sum(dim=(2, 3, 4)), rtol=1e-4, atol=2e-4 * scale * ref[0, 0].numel() ** 0.5)
    y_plain, _ = pointflow.conv3d_k3(xd, w.to(dev), stride, False)          # no affine: the plain convolution still
    assert _maxabs(y_plain, F.conv3d(x.double(), w.double(), None, stride, 1)) < 4e-6 * scale * max(1.0, (27 * Cin / 256.0) ** 0.5) * 3
    pointflow.flush_counters()
    assert _lib.status() == 0


@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("affine", ["rows", "lazy"])
@pytest.mark.parametrize("form", ["", "MFMA"])
def test_deconv3d_k3s2_applies_pending_batchnorm_before_the_skip_add(dev, skip, affine, form, monkeypatch):
    """pf_deconv3d_k3s2_f32 with the BatchNorm + ReLU of xa pending: y = deconv(relu(bn(xa)) + xb); both forms of the
    kernel (the shape rule would pick the lane-per-cell one here; PF_DECONV_MFMA=1 forces the matrix-core one)."""
    if form:
        monkeypatch.setenv("PF_DECONV_" + form, "1")
    gen = torch.Generator().manual_seed(7 + int(skip))
    N, Cin, Cout, D, H, W = 1, 32, 16, 6, 8, 10
    conv = torch.nn.ConvTranspose3d(Cin, Cout, 3, stride=2, padding=1, output_padding=1, bias=False)
    xa = torch.randn(N, Cin, D, H, W, generator=gen)
    xb = torch.randn(N, Cin, D, H, W, generator=gen) if skip else None
    xd = xa.to(dev)
    if affine == "rows":
        sc = torch.rand(1, Cin, generator=gen) + 0.5
        sh = torch.randn(1, Cin, generator=gen) * 0.3
        act = torch.relu(xa.double() * sc.double().view(1, Cin, 1, 1, 1) + sh.double().view(1, Cin, 1, 1, 1))
        aff = (sc.to(dev), sh.to(dev))
    else:
        bn = torch.nn.BatchNorm3d(Cin).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, Cin))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, Cin))
        T = 4
        part = torch.zeros((N, T, Cin, 2), dtype=torch.float64, device=dev)
        for t, ch in enumerate(torch.chunk(xd.double().reshape(N, Cin, -1), T, dim=2)):
            part[:, t, :, 0] = ch.sum(dim=2)
            part[:, t, :, 1] = (ch * ch).sum(dim=2)
        aff = pointflow.bn_affine_rows(xd, bn, N, part, lazy=True)
        assert isinstance(aff, pointflow.LazyAffine)
        act = torch.relu(F.batch_norm(xa.double(), None, None, bn.weight.double().cpu(), bn.bias.double().cpu(), True, 0.0,
                                      bn.eps))
    ref = F.conv_transpose3d(act + (xb.double() if skip else 0.0), conv.weight.double(), None, 2, 1, 1)
    y, part_y = pointflow.deconv3d_k3s2(xd, xb.to(dev) if skip else None, conv.weight.to(dev), True, in_affine=aff,
                                        samples_per_stat=N)
    scale = float(ref.abs().max())
    assert _maxabs(y, ref) < (2e-5 if affine == "lazy" else 4e-6) * scale
    sums = part_y.sum(dim=1).cpu()
    assert torch.allclose(sums[..., 0], ref.sum(dim=(2, 3, 4)), rtol=1e-4, atol=2e-4 * scale * ref[0, 0].numel() ** 0.5)
    pointflow.flush_counters()
    assert _lib.status() == 0


@pytest.mark.parametrize("N,C,D,H,W,T1,T2", [(1, 8, 6, 8, 20, 5, 3), (2, 8, 3, 5, 7, 4, 4)])
def test_batch_norm_act2_two_batchnorms_and_their_sum_in_one_pass(dev, N, C, D, H, W, T1, T2):
    gen = torch.Generator().manual_seed(N + D * H * W)
    x1 = torch.randn(N, C, D, H, W, generator=gen) * 1.5 + 0.3
    x2 = torch.randn(N, C, D, H, W, generator=gen) * 0.7 - 0.2
    mods, refs = [], []
    for seed in (1, 2):
        m = torch.nn.BatchNorm3d(C).to(dev).train()
        r = torch.nn.BatchNorm3d(C).double().train()
        with torch.no_grad():
            for b in (m, r):
                b.weight.copy_(torch.linspace(0.5, 1.5, C) * seed)
                b.bias.copy_(torch.linspace(-0.3, 0.3, C))
        mods.append(m)
        refs.append(r)

    def partials(x, T):
        xd = x.to(dev).double().reshape(N, C, -1)
        part = torch.zeros((N, T, C, 2), dtype=torch.float64, device=dev)
        for t, ch in enumerate(torch.chunk(xd, T, dim=2)):
            part[:, t, :, 0] = ch.sum(dim=2)
            part[:, t, :, 1] = (ch * ch).sum(dim=2)
        return part
    want = torch.relu(refs[1](x2.double())) + torch.relu(refs[0](x1.double()))
    a = x1.to(dev).clone()
    out = pointflow.batch_norm_act2_(a, mods[0], partials(x1, T1), x2.to(dev), mods[1], partials(x2, T2), N)
    pointflow.flush_counters()
    assert out.data_ptr() == a.data_ptr()
    assert _maxabs(out, want.detach()) < 2e-5 * float(want.abs().max())
    for m, r in zip(mods, refs):
        assert torch.allclose(m.running_mean.double().cpu(), r.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(m.running_var.double().cpu(), r.running_var, rtol=1e-5, atol=1e-6)
        assert int(m.num_batches_tracked) == 1
    assert _lib.status() == 0


# ---------------------------------------------------------------------------------------------
# the operator modules' plain forward (what the reference's model.py calls) reaches the HIP kernels when no autograd
# graph is needed, and equals the stock ATen composition the same module runs under autograd
# ---------------------------------------------------------------------------------------------
def _twin(mod, dev):
    import copy
    a = mod.to(dev).train()
    return a, copy.deepcopy(a)


def _buffers_close(a, b, rtol=1e-5, atol=1e-6):
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        if "num_batches" in k:
            assert int(x) == int(y), k
        elif "running" in k:
            assert torch.allclose(x, y, rtol=rtol, atol=atol), k


def test_image_conv_forward_takes_the_hip_route_without_autograd(dev):
    from pointmvsnet_amd.networks import ImageConv
    mod = ImageConv(8)
    synthetic.seed_weights(mod, seed=4)
    hip, stock = _twin(mod, dev)
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        got = hip(x)
    want = stock(x.clone().requires_grad_(True))                       # autograd on: the ATen composition
    assert set(got) == {"conv0", "conv1", "conv2", "conv3"}
    for k in got:
        scale = float(want[k].abs().max())
        err = _maxabs(got[k], want[k].detach())
        report("image_conv_forward_hip_" + k, err=err, scale=scale)
        assert got[k].shape == want[k].shape and err < 3e-5 * scale
        assert not got[k].requires_grad
    _buffers_close(hip, stock, rtol=1e-4, atol=1e-5)


def test_volume_conv_forward_takes_the_hip_route_without_autograd(dev):
    from pointmvsnet_amd.networks import VolumeConv
    mod = VolumeConv(64, 8)
    synthetic.seed_weights(mod, seed=2)
    hip, stock = _twin(mod, dev)
    x = torch.rand(1, 64, 16, 32, 40, generator=torch.Generator().manual_seed(9)).to(dev)
    with torch.no_grad():
        got = hip(x)
    want = stock(x.clone().requires_grad_(True)).detach()
    err, scale = _maxabs(got, want), float(want.abs().max())
    report("volume_conv_forward_hip", err=err, scale=scale)
    assert got.shape == want.shape and err < 2e-5 * scale
    _buffers_close(hip, stock, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,C,N,widths", [(1, 224, 25600, (64, 64, 16)), (2, 224, 1000, (64, 64, 16)), (3, 40, 77, (32, 128))])
def test_shared_mlp_forward_takes_the_hip_route_without_autograd(dev, B, C, N, widths):
    from pointmvsnet_amd.nn.mlp import SharedMLP
    mod = SharedMLP(C, widths)
    synthetic.seed_weights(mod, seed=6)
    hip, stock = _twin(mod, dev)
    x = torch.randn(B, C, N, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        got = hip(x)
    want = stock(x.clone().requires_grad_(True)).detach()
    err, scale = _maxabs(got, want), float(want.abs().max())
    report("shared_mlp_forward_hip_%d_%d" % (B, N), err=err, scale=scale)
    assert got.shape == want.shape and got.is_contiguous() and err < 2e-5 * scale
    _buffers_close(hip, stock, rtol=1e-4, atol=1e-5)
