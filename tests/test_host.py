"""CPU: host-side logic -- module surface, state-dict contract, camera packing, compat aliasing, loud
failure without a GPU, synthetic-input determinism."""
import json
import os

import pytest
import torch

from conftest import GOLDEN_DIR, REFERENCE_DIR
from pointmvsnet_amd import synthetic
from pointmvsnet_amd.model import PointMVSNet, PointMVSNetLoss, PointMVSNetMetric, _Cameras


def test_state_dict_keys_match_reference_checkpoint_contract():
    want = json.load(open(os.path.join(GOLDEN_DIR, "state_dict_keys.json")))
    got = {k: list(v.shape) for k, v in PointMVSNet().state_dict().items()}
    assert got == want
    assert sum(p.numel() for p in PointMVSNet().parameters()) == 698936


def test_seed_weights_is_construction_order_independent():
    a, b = PointMVSNet(), PointMVSNet()
    synthetic.seed_weights(a, 0)
    synthetic.seed_weights(b, 0)
    for (ka, va), (kb, vb) in zip(sorted(a.state_dict().items()), sorted(b.state_dict().items())):
        assert ka == kb and torch.equal(va, vb)
    c = PointMVSNet()
    synthetic.seed_weights(c, 1)
    assert not torch.equal(c.state_dict()["flow_mlp.1.weight"], a.state_dict()["flow_mlp.1.weight"])


def test_synthetic_scene_geometry_is_sane():
    data, scales, inters = synthetic.make_config("cfg2")
    assert data["img_list"].shape == (1, 3, 3, 512, 640) and data["cam_params_list"].shape == (1, 3, 2, 4, 4)
    R = data["cam_params_list"][0, :, 0, :3, :3].double()
    eye = torch.eye(3, dtype=torch.float64).expand(3, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-6)             # proper rotations
    again, _, _ = synthetic.make_config("cfg2")
    assert torch.equal(again["img_list"], data["img_list"])                  # deterministic
    other, _, _ = synthetic.make_config("cfg2", seed=1)
    assert not torch.equal(other["img_list"], data["img_list"])


def test_camera_pack_layout_matches_header():
    data, _, _ = synthetic.make_config("tiny")
    cam = _Cameras(data["cam_params_list"], True)
    K = cam.flow_intrinsics(0.25)
    interval = 0.75 * cam.depth_interval
    pack = cam.packed(K, data["mean"], data["std"], interval)
    V = 3
    assert pack.shape == (1, 27 + 21 * V + 1)           # PF_CAM_FLOATS(V) + the hypothesis interval
    assert torch.equal(pack[0, -1], interval[0])
    assert torch.allclose(pack[0, 0:9].view(3, 3) @ K[0, 0], torch.eye(3), atol=1e-4)
    assert torch.equal(pack[0, 18:21], cam.t[0, 0].reshape(-1))
    assert torch.equal(pack[0, 21:24], data["mean"][0]) and torch.equal(pack[0, 24:27], data["std"][0])
    for v in range(V):
        base = 27 + 21 * v
        assert torch.equal(pack[0, base:base + 9].view(3, 3), K[0, v])
        assert torch.equal(pack[0, base + 9:base + 21].view(3, 4), cam.ext[0, v])
    # intrinsic scaling rules of reference model.py:59-61 and :162-163
    raw = data["cam_params_list"][0, 0, 1, 0, 0]
    assert torch.isclose(cam.K_coarse[0, 0, 0, 0], raw / 8.0)
    assert torch.isclose(_Cameras(data["cam_params_list"], False).K_coarse[0, 0, 0, 0], raw / 2.0)
    assert torch.isclose(_Cameras(data["cam_params_list"], False).flow_intrinsics(0.25)[0, 0, 0, 0], raw)


def test_scene_plan_blocks_are_one_aligned_buffer():
    from pointmvsnet_amd.model import ScenePlan
    data, scales, inters = synthetic.make_config("tiny")
    plan = ScenePlan(torch.device("cpu"), 1, 3, 128, 192, scales, inters, True, 8).update_(data)
    cam = _Cameras(data["cam_params_list"], True)
    assert torch.equal(plan.d("K_coarse"), cam.K_coarse) and torch.equal(plan.d("ext"), cam.ext)
    assert torch.equal(plan.d("depths")[0], torch.linspace(425.0, float(cam.depth_end[0]), 8))
    assert torch.equal(plan.d("sa_params")[0], torch.stack([cam.depth_start[0], cam.depth_end[0], cam.depth_interval[0]]))
    assert torch.equal(plan.d("pack1")[0, -1], (0.75 * cam.depth_interval)[0])
    for name in ("K_coarse", "ext", "Kinv0", "Rinv0", "t0", "depths", "sa_params", "pack0", "pack1"):
        assert plan.d(name).data_ptr() % 16 == 0
    assert plan.matches(torch.device("cpu"), 1, 3, 128, 192, scales, inters, True, 8)
    assert not plan.matches(torch.device("cpu"), 1, 3, 128, 192, scales, inters, False, 8)
    other, _, _ = synthetic.make_config("tiny", seed=4)
    plan.update_(other)
    assert torch.equal(plan.d("ext"), _Cameras(other["cam_params_list"], True).ext)


def test_train_plan_holds_every_host_constant_of_the_autograd_forward():
    """TrainPlan (the constants of the training forward in one block, what makes the step capturable): the same
    values the reference's forward derives on the host (model.py:59-61, :87-100, :162-178), train-mode intrinsics."""
    from pointmvsnet_amd.model import TrainPlan
    data, scales, inters = synthetic.make_config("tiny", train_intrinsics=True)
    plan = TrainPlan(torch.device("cpu"), 1, 3, 8, scales, inters, False).update_(data)
    cam = _Cameras(data["cam_params_list"], False)
    assert torch.equal(plan.d("K_coarse"), cam.K_coarse) and torch.equal(plan.d("ext"), cam.ext)
    assert torch.equal(plan.d("Kinv0")[:, 0], torch.inverse(cam.K_coarse[:, 0]))
    assert torch.equal(plan.d("Rinv0"), cam.R_inv[:, 0:1]) and torch.equal(plan.d("t0"), cam.t[:, 0:1])
    assert torch.equal(plan.d("depths")[0], torch.linspace(float(cam.depth_start[0]), float(cam.depth_end[0]), 8))
    assert torch.equal(plan.d("d_start"), cam.depth_start) and torch.equal(plan.d("d_int"), cam.depth_interval)
    assert torch.equal(plan.d("mean")[:, :, 0], data["mean"].float()) and torch.equal(plan.d("std")[:, :, 0], data["std"].float())
    for i, (s_, inter) in enumerate(zip(scales, inters)):
        K_flow = cam.flow_intrinsics(s_)
        assert torch.equal(plan.d("K_flow%d" % i), K_flow)
        assert torch.equal(plan.d("Kinv_flow%d" % i)[:, 0], torch.inverse(K_flow[:, 0]))
        assert torch.equal(plan.d("interval%d" % i), inter * cam.depth_interval)
        for name in ("K_flow%d" % i, "Kinv_flow%d" % i, "interval%d" % i):
            assert plan.d(name).data_ptr() % 16 == 0
    assert plan.matches(torch.device("cpu"), 1, 3, 8, scales, inters, False)
    assert not plan.matches(torch.device("cpu"), 1, 3, 8, scales, inters, True)
    other, _, _ = synthetic.make_config("tiny", seed=4, train_intrinsics=True)
    plan.update_(other)
    assert torch.equal(plan.d("ext"), _Cameras(other["cam_params_list"], False).ext)


def test_no_pack_cache_context_bypasses_and_restores_the_cache():
    from pointmvsnet_amd import pointflow
    w = torch.nn.Parameter(torch.randn(8, 4, 1))
    a, _ = pointflow.pack_weight_t(w)
    b, _ = pointflow.pack_weight_t(w)
    assert a is b                                           # cached
    with pointflow.no_pack_cache():
        c, _ = pointflow.pack_weight_t(w)
        with pointflow.no_pack_cache():
            d, _ = pointflow.pack_weight_t(w)
        e, _ = pointflow.pack_weight_t(w)
    assert c is not a and d is not c and e is not d and torch.equal(c, a)
    f, _ = pointflow.pack_weight_t(w)
    assert f is a                                           # the cache itself was not touched


def test_operators_fail_loudly_without_gpu():
    from pointmvsnet_amd.functions.gather_knn import gather_knn
    from pointmvsnet_amd.networks import EdgeConv
    from pointmvsnet_amd.utils.feature_fetcher import FeatureFetcher
    from pointmvsnet_amd.utils.torch_utils import get_knn_3d
    with pytest.raises(RuntimeError, match="no CPU path"):
        gather_knn(torch.randn(1, 2, 4), torch.zeros(1, 4, 2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        get_knn_3d(torch.randn(1, 3, 5, 4, 4), 5, knn=16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        FeatureFetcher()(torch.randn(1, 2, 4, 8, 8), torch.randn(1, 3, 5), torch.eye(3).expand(1, 2, 3, 3), None)
    with pytest.raises(RuntimeError, match="no CPU path"):
        EdgeConv(8, 32)(torch.randn(1, 8, 10), torch.zeros(1, 10, 4, dtype=torch.int64))
    data, s, i = synthetic.make_config("tiny")
    with pytest.raises(RuntimeError, match="GPU"):
        PointMVSNet()(data, s, i, isFlow=True, isTest=True)


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    from pointmvsnet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _lib.load()


def test_loss_and_metric_surface():
    loss_fn, metric_fn = PointMVSNetLoss(valid_threshold=8.0), PointMVSNetMetric(valid_threshold=8.0)
    g = torch.Generator().manual_seed(0)
    gt = 500 + 50 * torch.rand(2, 1, 32, 40, generator=g)
    gt[:, :, :4] = 0.0
    cams = torch.zeros(2, 3, 2, 4, 4)
    cams[:, 0, 1, 3, 1] = 10.6
    preds = {"coarse_depth_map": gt[:, :, ::2, ::2] + 3.0, "flow1": gt[:, :, ::2, ::2] + 1.0, "flow2": gt + 0.5}
    labels = {"gt_depth_img": gt, "cam_params_list": cams}
    losses = loss_fn(preds, labels, True)
    assert set(losses) == {"coarse_loss", "flow1_loss", "flow2_loss"}
    # masked MAE in interval units, summed over the batch of 2, divided by the 3 terms
    assert torch.isclose(losses["coarse_loss"], torch.tensor(2 * (3.0 / 10.6) / 3), rtol=1e-4)
    assert torch.isclose(losses["flow2_loss"], torch.tensor(2 * (0.5 / (0.375 * 10.6)) / 3), rtol=1e-4)
    m = metric_fn(preds, labels, True)
    assert float(m["<1_pct_cor"]) == 1.0 and float(m["<1_pct_flow2"]) == 1.0
    assert set(m) == {"<1_pct_cor", "<3_pct_cor", "<1_pct_flow1", "<3_pct_flow1", "<1_pct_flow2", "<3_pct_flow2"}


@pytest.mark.skipif(not os.path.isdir(REFERENCE_DIR), reason="reference tree only exists in the build container")
def test_reference_model_py_imports_unchanged_on_our_operator_layer():
    from pointmvsnet_amd import compat, networks
    ref = compat.load_reference_model(os.path.join(REFERENCE_DIR, "pointmvsnet", "model.py"))
    net = ref.PointMVSNet()
    assert isinstance(net.flow_edge_conv[1], networks.EdgeConv)              # our classes behind their names
    assert type(net.feature_fetcher).__module__ == "pointmvsnet_amd.utils.feature_fetcher"
    want = json.load(open(os.path.join(GOLDEN_DIR, "state_dict_keys.json")))
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == want
    assert ref.get_knn_3d.__module__ == "pointmvsnet_amd.utils.torch_utils"


def test_pack_cache_follows_data_swaps_and_pins():
    """``param.data = ...`` / ``module.double()`` do not bump ``_version``; the cache must still miss."""
    from pointmvsnet_amd import pointflow as pf
    w = torch.nn.Parameter(torch.randn(16, 8, 1))
    a, _ = pf.pack_weight_t(w)
    assert pf.pack_weight_t(w)[0] is a
    w.data = torch.randn(16, 8, 1)                        # version counter unchanged, storage moved
    b, _ = pf.pack_weight_t(w)
    assert b is not a and torch.equal(b[:, :16], w.detach()[:, :, 0].t())
    with torch.no_grad():
        w.add_(1.0)                                       # in-place update: version counter
    c, _ = pf.pack_weight_t(w)
    assert c is not b and torch.equal(c[:, :16], w.detach()[:, :, 0].t())
    conv = torch.nn.Conv2d(8, 16, 3, bias=False)
    p1 = pf.pack_conv2d_wide_weight(conv.weight)
    conv.double()                                         # dtype change through .data
    p2 = pf.pack_conv2d_wide_weight(conv.weight)
    assert p2 is not p1 and p2.dtype == torch.float32
    # pinned entries survive eviction pressure; unpinned least-recently-used ones go first
    pf.pack_log_begin()
    keep = torch.nn.Parameter(torch.randn(4, 4, 1))
    pk, _ = pf.pack_weight_t(keep)
    pins = pf.pack_log_end(pin=True)
    junk = [torch.nn.Parameter(torch.randn(4, 4, 1)) for _ in range(300)]
    for j in junk:
        pf.pack_weight_t(j)
    assert pf.pack_weight_t(keep)[0] is pk and len(pf._pack_cache) <= pf._PACK_CAP + 1
    assert not pf.pack_entries_stale(pins)
    keep.data = torch.zeros(4, 4, 1)
    assert pf.pack_entries_stale(pins)
    pf.pack_unpin(pins)


def test_deferred_batchnorm_jobs_are_layered_by_module_call_order(monkeypatch):
    """pointflow.flush_lazy_stats: the jobs of one finalize launch run concurrently, so two deferred jobs of the SAME
    BatchNorm module (the flow MLP is called once per PointFlow iteration) must go to consecutive launches, in call
    order -- a shared launch would apply only one of the two running-statistics updates."""
    import contextlib
    from pointmvsnet_amd import _lib, pointflow

    launches = []
    monkeypatch.setattr(pointflow, "bn_finalize_jobs", lambda jobs: launches.append([j.T for j in jobs]))
    monkeypatch.setattr(pointflow.torch.cuda, "device", lambda d: contextlib.nullcontext())

    def lazy(tag, running_mean_ptr):
        job = _lib.BnJob(1, tag, 1, 0, 1, 1.0, 1.0, 2, 3, running_mean_ptr, running_mean_ptr, 0.1, 1e-5, 1, 1, 4, 5, 1)
        z = pointflow.LazyAffine(job, (), torch.zeros(1), torch.zeros(1))
        z.job_ptr()                                             # queues it (device key "cpu")
        return z

    a1, b1, c1 = lazy(11, 1000), lazy(21, 2000), lazy(31, None)   # three modules (one without running statistics)
    a2, b2 = lazy(12, 1000), lazy(22, 2000)                       # the first two again (second iteration)
    a3 = lazy(13, 1000)
    done = lazy(99, 3000)
    done.done = True                                              # materialised meanwhile: must not be launched again
    pointflow.flush_lazy_stats()
    assert launches == [[11, 21, 31], [12, 22], [13]]
    assert all(z.done for z in (a1, b1, c1, a2, b2, a3))
    launches.clear()
    pointflow.flush_lazy_stats()                                  # nothing left
    assert launches == []


@pytest.mark.parametrize("cfg,batch", [("cfg2", 1), ("cfg3", 1), ("cfg5", 1), ("tiny", 2)])
@pytest.mark.parametrize("is_test", [True, False])
def test_scene_plan_block_equals_the_reference_camera_algebra_bit_for_bit(cfg, batch, is_test):
    """ScenePlan.fill_host_ (a dozen NumPy / LAPACK operations per scene) against the ``_Cameras`` composition, which
    issues the reference's own ATen-CPU calls (reference model.py:54-61, :159-170): every float of the block equal."""
    from pointmvsnet_amd.model import ScenePlan, _Cameras, _host_cams
    for seed in (0, 3):
        if batch == 1:
            data, scales, inters = synthetic.make_config(cfg, seed=seed)
        else:
            data, scales, inters = synthetic.make_scene(128, 192, 3, 8, seed=11 + seed, batch=batch), (0.125, 0.25), (1.0, 0.75)
        B, V, _, H, W = data["img_list"].shape
        D = int(data["cam_params_list"][0, 0, 1, 3, 2])
        plan = ScenePlan(torch.device("cpu"), B, V, H, W, scales, inters, is_test, D).fill_host_(data)
        cam = _Cameras(_host_cams(data), is_test)
        want = {"K_coarse": cam.K_coarse, "ext": cam.ext, "Kinv0": torch.inverse(cam.K_coarse[:, 0]),
                "Rinv0": cam.R_inv[:, 0], "t0": cam.t[:, 0],
                "depths": torch.stack([torch.linspace(float(cam.depth_start[b]), float(cam.depth_end[b]), D)
                                       for b in range(B)]),
                "sa_params": torch.stack([cam.depth_start, cam.depth_end, cam.depth_interval], dim=1)}
        for i, (s, inter) in enumerate(zip(scales, inters)):
            want["pack%d" % i] = cam.packed(cam.flow_intrinsics(s), data["mean"], data["std"], inter * cam.depth_interval)
        for name, t in want.items():
            got = plan._h(name)
            assert torch.equal(got, t.reshape(got.shape).float()), (cfg, seed, is_test, name)


def test_flat_rmsprop_state_is_interchangeable_with_torch_rmsprop():
    """train_step.FlatRMSprop is a torch.optim.Optimizer on the reference's two parameter groups (solver.py:33-52:
    everything but '.bn.' with weight decay, then the '.bn.' parameters) and its state_dict is torch.optim.RMSprop's
    layout, both directions -- host logic only (step() is a HIP launch, tests/test_gpu_model.py)."""
    from pointmvsnet_amd import distributed
    from pointmvsnet_amd.model import PointMVSNet
    from pointmvsnet_amd.train_step import FlatRMSprop, param_groups
    net = PointMVSNet()
    bucket = distributed.GradBucket(net)
    flat = FlatRMSprop(bucket, list(net.named_parameters()), lr=1e-3, alpha=0.9, weight_decay=1e-4)
    ref = torch.optim.RMSprop(param_groups(net, 1e-4), lr=1e-3, alpha=0.9)
    assert isinstance(flat, torch.optim.Optimizer) and flat.attached() and bucket.attached()
    assert [len(g["params"]) for g in flat.param_groups] == [len(g["params"]) for g in ref.param_groups]
    for g, h in zip(flat.param_groups, ref.param_groups):
        assert all(a is b for a, b in zip(g["params"], h["params"])) and g["weight_decay"] == h["weight_decay"]
    for p in net.parameters():
        p.grad.add_(0.01)
    ref.step()                                              # torch builds its state; ours must accept it
    flat.load_state_dict(ref.state_dict())
    sd, sd_ref = flat.state_dict(), ref.state_dict()
    assert flat.steps == 1 and sorted(sd["state"]) == sorted(sd_ref["state"])
    assert all(torch.equal(sd["state"][i]["square_avg"], sd_ref["state"][i]["square_avg"]) for i in sd_ref["state"])
    assert [g["params"] for g in sd["param_groups"]] == [g["params"] for g in sd_ref["param_groups"]]
    ref.load_state_dict(sd)                                 # ... and torch accepts ours
    assert flat.wd is not None and float(flat.wd.sum()) > 0
    # the '.bn.' parameters carry no decay: their span of the per-element vector is zero
    for name, p in net.named_parameters():
        o, n = flat._span[id(p)]
        assert float(flat.wd[o:o + n].abs().sum()) == (0.0 if ".bn." in name else float(flat.wd[o:o + n].abs().sum()))
        if ".bn." in name:
            assert float(flat.wd[o:o + n].abs().sum()) == 0.0
    sched = torch.optim.lr_scheduler.StepLR(flat, step_size=1, gamma=0.1)     # the reference's scheduler attaches
    assert flat.lr == 1e-3
    flat.set_lr(5e-4)
    assert flat.lr == 5e-4 and sched is not None
    flat.zero_grad(set_to_none=True)                        # must keep the bucket views
    assert bucket.attached() and float(bucket.flat.abs().sum()) == 0.0
    # a frozen parameter would shift torch's state indices against ours: refused with a message, not a size mismatch later
    net2 = PointMVSNet()
    next(net2.parameters()).requires_grad_(False)
    with pytest.raises(ValueError, match="trainable"):
        FlatRMSprop(distributed.GradBucket(net2), list(net2.named_parameters()))
    flat.param_groups[0]["maximize"] = True
    with pytest.raises(NotImplementedError):
        flat._hyper()
