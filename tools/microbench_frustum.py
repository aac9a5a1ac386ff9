"""Micro-benchmark of the coarse warp (pf_frustum_variance_cl_f32) on the BASELINE shapes (cfg 2 / 3 / 5)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import synthetic  # noqa: E402
from pointmvsnet_amd.model import ScenePlan  # noqa: E402
from pointmvsnet_amd.utils.feature_fetcher import ChannelLast, frustum_variance  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for cfg in ("cfg2", "cfg3", "cfg5"):
    data, scales, inters = synthetic.make_config(cfg)
    B, V, _, H, W = data["img_list"].shape
    D = int(data["cam_params_list"][0, 0, 1, 3, 2])
    plan = ScenePlan(dev, B, V, H, W, scales, inters, True, D).update_(data)
    maps = ChannelLast(torch.randn(B, V, H // 8, W // 8, 64, device=dev))
    args = (maps, plan.d("Kinv0"), plan.d("Rinv0"), plan.d("t0"), plan.d("depths"), plan.d("K_coarse"), plan.d("ext"))
    for _ in range(5):
        cost, world = frustum_variance(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        frustum_variance(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    mb = cost.numel() * 4 / 1e6
    print("%s V=%d D=%d %dx%d: %.1f us, cost volume %.1f MB -> %.2f TB/s, checksum %.9e" % (
        cfg, V, D, H // 8, W // 8, us, mb, mb / us, float(cost.double().sum())), flush=True)
