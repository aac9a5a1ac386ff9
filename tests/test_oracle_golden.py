"""CPU: pin the oracle against (a) vectors produced by running the reference itself
(tests/golden/make_golden.py), (b) the reference's own two known-answer self tests, and
(c) independent first-principles NumPy versions (oracle/bruteforce.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import bruteforce as BF
from oracle import pointflow_oracle as O
from pointmvsnet_amd import synthetic
from pointmvsnet_amd.model import PointMVSNet


def test_gather_selftest_matches_reference_known_answer():
    # reference functions/gather_knn.py:27-56
    g = load_golden("gather_knn_selftest")
    out = O.gather_knn(g["feature"], g["index"])
    assert torch.equal(out, g["out"])
    gi = O.gather_knn_backward(torch.ones_like(out), g["index"])
    assert torch.allclose(gi, g["grad"])
    assert np.array_equal(BF.gather(g["feature"].numpy(), g["index"].numpy()), g["out"].numpy())
    assert np.allclose(BF.scatter_add(np.ones_like(g["out"].numpy()), g["index"].numpy()), g["grad"].numpy())


def test_fetch_selftest_exact_texel():
    # reference utils/feature_fetcher.py:63-97: the point projecting to pixel (60.5, 80.5) fetches
    # feature[..., 80, 60] (rtol 1e-2 in the reference's own check)
    g = load_golden("feature_fetch_selftest")
    torch.manual_seed(0)
    E = torch.rand(3, 2, 3, 4)
    feats = torch.rand(3, 2, 16, 240, 320)
    assert torch.equal(E, g["E"])
    out = O.fetch_features(feats, g["pts"], g["K"], g["E"])
    assert torch.equal(out, g["out"])
    # random (non-rotation) extrinsics make the inverse ill-conditioned, so the projection lands within
    # ~1e-4 px of the texel centre: the reference's rtol 1e-2 plus an absolute floor for near-zero texels
    assert np.allclose(out[:, 0, :, 0].numpy(), g["expected_view0"].numpy(), rtol=1e-2, atol=1e-4)
    assert np.allclose(out[:, 0, :, 0].numpy(), feats[:, 0, :, 80, 60].numpy(), rtol=1e-2, atol=1e-4)


def test_fetch_random_vs_reference_and_first_principles():
    g = load_golden("feature_fetch_random")
    out = O.fetch_features(g["feats"], g["pts"], g["K"], g["E"])
    assert torch.equal(out, g["out"])
    bf = BF.fetch_bilinear(g["feats"].numpy(), g["pts"].numpy(), g["K"].numpy(), g["E"].numpy())
    scale = float(g["feats"].abs().max())
    assert np.abs(bf - out.numpy()).max() < 2e-4 * scale      # float32 projection vs float64 definition
    assert (out == 0).any() and (out != 0).any()             # both inside and outside the maps


@pytest.mark.parametrize("name,ks,knn", [("knn_lattice_far", 5, 16), ("knn_lattice_origin", 5, 16),
                                         ("knn_lattice_k3", 3, 8)])
def test_knn_vs_reference_and_first_principles(name, ks, knn):
    g = load_golden(name)
    idx, code = O.knn_lattice(g["xyz"], ks, knn, return_code=True)
    assert torch.equal(idx, g["idx"])                         # same ATen calls -> same tie order too
    for b in range(g["xyz"].shape[0]):
        bf_idx, bf_code = BF.knn_window(g["xyz"][b].numpy(), ks, knn)
        d2 = BF.knn_window_d2(g["xyz"][b].numpy(), ks)
        assert np.array_equal(d2, O.knn_lattice_d2(g["xyz"][b:b + 1], ks)[0].numpy())   # bit-exact distances
        oc = code[b].numpy()
        differ = np.where((np.sort(bf_code, axis=1) != np.sort(oc, axis=1)).any(axis=1))[0]
        # candidate sets may differ only inside tie groups: identical ranked distances
        for n in differ:
            assert np.array_equal(np.sort(d2[bf_code[n], n]), np.sort(d2[oc[n], n]))
        if name != "knn_lattice_origin":
            assert len(differ) == 0
        same_idx = (np.sort(bf_idx, axis=1) == np.sort(idx[b].numpy(), axis=1)).all(axis=1)
        assert same_idx[np.setdiff1d(np.arange(len(same_idx)), differ)].all()


def test_knn_origin_fixture_exercises_padding_and_clamp():
    g = load_golden("knn_lattice_origin")
    xyz = g["xyz"][0].numpy()
    _, code = BF.knn_window(xyz, 5, 16)
    D, H, W = xyz.shape[1:]
    n = np.arange(D * H * W)
    d, h, w = n // (H * W), (n // W) % H, n % W
    dd, dh, dw = code // 25 - 2, (code % 25) // 5 - 2, code % 5 - 2
    outside = ((d[:, None] + dd < 0) | (d[:, None] + dd >= D) | (h[:, None] + dh < 0) | (h[:, None] + dh >= H)
               | (w[:, None] + dw < 0) | (w[:, None] + dw >= W))
    assert outside.any(), "fixture must contain picks of zero-padded candidates"


def test_knn_strided_view():
    g = load_golden("knn_lattice_strided")
    sub = g["xyz_full"].view(1, 3, 5, 6, 2, 8, 2)[:, :, :, :, 1, :, 0]
    assert not sub.is_contiguous()
    assert torch.equal(O.knn_lattice(sub, 5, 16), g["idx"])


@pytest.mark.parametrize("name,prefix_concat", [("edgeconv_noc", False), ("edgeconv_32", True), ("edgeconv_64", True)])
def test_edgeconv_vs_reference_and_first_principles(name, prefix_concat):
    from pointmvsnet_amd.networks import EdgeConv, EdgeConvNoC
    g = load_golden(name)
    cin = g["x"].shape[1]
    cout = g["y"].shape[1] // (2 if prefix_concat else 1)
    mod = (EdgeConv if prefix_concat else EdgeConvNoC)(cin, cout)
    synthetic.seed_weights(mod, seed=1)
    sd = {"m." + k: v for k, v in mod.state_dict().items()}
    track = {}
    y = O.edge_conv(g["x"], g["idx"], sd, "m", prefix_concat, track)
    assert torch.equal(y, g["y"])
    assert torch.equal(track["m.bn.running_mean"], g["running_mean"])
    assert torch.equal(track["m.bn.running_var"], g["running_var"])
    bf, (mean, var_unb) = BF.edge_conv(g["x"].numpy(), g["idx"].numpy(), sd["m.conv1.weight"][:, :, 0].numpy(),
                                       sd["m.conv2.weight"][:, :, 0].numpy(), sd["m.bn.weight"].numpy(),
                                       sd["m.bn.bias"].numpy(), prefix_concat)
    assert np.abs(bf - y.numpy()).max() < 5e-5
    rm0, rv0 = mod.bn.running_mean.numpy(), mod.bn.running_var.numpy()
    assert np.allclose(0.9 * rm0 + 0.1 * mean, g["running_mean"].numpy(), atol=1e-5)
    assert np.allclose(0.9 * rv0 + 0.1 * var_unb, g["running_var"].numpy(), rtol=1e-4)


def test_volume_conv_vs_reference():
    from pointmvsnet_amd.networks import VolumeConv
    g = load_golden("volume_conv")
    mod = VolumeConv(64, 8)
    synthetic.seed_weights(mod, seed=2)
    sd = {"v." + k: v for k, v in mod.state_dict().items()}
    y = O.volume_conv(g["x"], sd, "v")
    assert torch.equal(y, g["y"])
    mod.train()
    with torch.no_grad():
        assert torch.equal(mod(g["x"]), g["y"])               # our nn.Module == reference module on CPU


@pytest.mark.parametrize("tag,cfg,is_test", [("model_tiny_test", "tiny", True), ("model_tiny_train", "tiny", False),
                                             ("model_small_test", "small", True),
                                             ("model_cfg4_train", "cfg4", False)])
def test_whole_forward_vs_reference(tag, cfg, is_test):
    g = load_golden(tag)
    data, img_scales, inter_scales = synthetic.make_config(cfg, train_intrinsics=not is_test)
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    track = {}
    with torch.no_grad():
        preds = O.forward(net.state_dict(), data, img_scales, inter_scales, True, is_test, track=track)
    for key, val in preds.items():
        if key == "world_points":
            assert torch.equal(val[:, :, :4096], g["world_points_head"])
        else:
            assert torch.equal(val, g[key]), key
    for key in g:
        if key.startswith("sd:"):
            assert torch.allclose(track[key[3:]].float(), g[key].float(), rtol=1e-6, atol=1e-7), key
