"""Micro-benchmark: the EdgeConv backward's passes at the two lattice sizes of BASELINE config 4 (25 600 and 102 400 points),
two walks (PF_EDGE_BWD_SUMS=1, default) or three; run under `rocprofv3 --kernel-trace --stats` for the per-kernel times,
or alone for the per-layer event times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import pointflow, synthetic  # noqa: E402
from pointmvsnet_amd.networks import EdgeConv, EdgeConvNoC  # noqa: E402
from pointmvsnet_amd.utils.torch_utils import get_knn_3d  # noqa: E402

dev = torch.device("cuda:0")


FLUSH = os.environ.get("MB_FLUSH", "0") != "0"     # evict the caches between repetitions (a 1 GB fill)
_junk = torch.empty((256 << 20,), dtype=torch.float32, device=dev) if FLUSH else None


def timeit(fn, reps=20):
    if FLUSH:
        inner = fn

        def fn():
            _junk.fill_(1.0)
            inner()
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


SIZES = {"small": (5, 64, 80), "big": (5, 128, 160)}
for D, H, W in [SIZES[a] for a in sys.argv[1:]] or list(SIZES.values()):
    N = D * H * W
    gen = torch.Generator().manual_seed(0)
    zs = torch.linspace(-0.2, 0.2, D).view(1, 1, D, 1, 1).expand(1, 1, D, H, W)
    ys = torch.linspace(-1.0, 1.0, H).view(1, 1, 1, H, 1).expand(1, 1, D, H, W)
    xs = torch.linspace(-1.25, 1.25, W).view(1, 1, 1, 1, W).expand(1, 1, D, H, W)
    xyz = (torch.cat([xs, ys, zs], 1) + 0.004 * torch.randn(1, 3, D, H, W, generator=gen)).contiguous().to(dev)
    idx = get_knn_3d(xyz, 5, knn=16)
    for cls, cin, cout in ((EdgeConvNoC, 136, 32), (EdgeConv, 32, 32), (EdgeConv, 64, 64)):
        mod = cls(cin, cout)
        synthetic.seed_weights(mod, 1)
        mod = mod.to(dev).train()
        x = torch.randn(1, cin, N, generator=gen).to(dev).requires_grad_(True)
        go = torch.randn(1, (2 if mod.concat else 1) * cout, N, generator=gen).to(dev)
        y = mod(x, idx)
        t = timeit(lambda: y.backward(go, retain_graph=True))
        print("N %6d %s %d->%d backward (index inversion cached): %.0f us  [PF_EDGE_BWD_SUMS=%s PF_EDGE_FINISH_DBG=%s]"
              % (N, cls.__name__, cin, cout, t, os.environ.get("PF_EDGE_BWD_SUMS", "1"),
                 os.environ.get("PF_EDGE_FINISH_DBG", "0")), flush=True)
