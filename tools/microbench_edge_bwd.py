"""Micro-benchmark: EdgeConv backward at BASELINE config 4's size (one 102 400-point lattice), the de rows by the
inverse-list gather (default) against the float-atomic scatter."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import pointflow, synthetic  # noqa: E402
from pointmvsnet_amd.networks import EdgeConv, EdgeConvNoC  # noqa: E402
from pointmvsnet_amd.utils.torch_utils import get_knn_3d  # noqa: E402

dev = torch.device("cuda:0")
D, H, W = 5, 128, 160
N = D * H * W
gen = torch.Generator().manual_seed(0)
zs = torch.linspace(-0.2, 0.2, D).view(1, 1, D, 1, 1).expand(1, 1, D, H, W)
ys = torch.linspace(-1.0, 1.0, H).view(1, 1, 1, H, 1).expand(1, 1, D, H, W)
xs = torch.linspace(-1.25, 1.25, W).view(1, 1, 1, 1, W).expand(1, 1, D, H, W)
xyz = (torch.cat([xs, ys, zs], 1) + 0.004 * torch.randn(1, 3, D, H, W, generator=gen)).contiguous().to(dev)
idx = get_knn_3d(xyz, 5, knn=16)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


for cls, cin, cout in ((EdgeConvNoC, 136, 32), (EdgeConv, 32, 32), (EdgeConv, 64, 64)):
    mod = cls(cin, cout)
    synthetic.seed_weights(mod, 1)
    mod = mod.to(dev).train()
    x = torch.randn(1, cin, N, generator=gen).to(dev).requires_grad_(True)
    go = torch.randn(1, (2 if mod.concat else 1) * cout, N, generator=gen).to(dev)
    y = mod(x, idx)
    for det in (True, False):
        pointflow.DETERMINISTIC_BACKWARD = det

        def bwd():
            pointflow._inverse_cache.clear()
            y.backward(go, retain_graph=True)
        t_all = timeit(bwd)

        def bwd_cached():
            y.backward(go, retain_graph=True)
        t_cached = timeit(bwd_cached)
        print("%s %d->%d backward: %s %.0f us (index inversion cached: %.0f us)"
              % (cls.__name__, cin, cout, "inverse-list gather" if det else "atomic scatter     ", t_all, t_cached), flush=True)
t_inv = timeit(lambda: (pointflow._inverse_cache.clear(), pointflow.knn_inverse(idx, 1, N, 16)))
print("pf_knn_inverse alone: %.0f us" % t_inv)
