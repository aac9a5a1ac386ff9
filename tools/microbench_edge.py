"""Micro-benchmark of the EdgeConv gather passes (pf_edge_stats_f32 / pf_edge_apply_f32) on the flow-2 lattice of
BASELINE config 2 (4 sub-grids x 5 x 64 x 80 points), window codes from the lattice kNN."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import _lib, pointflow  # noqa: E402
from pointmvsnet_amd.utils.torch_utils import knn_lattice  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


for G, hs, ws in ((4, 64, 80), (1, 64, 80), (16, 60, 80)):
    Ng = 5 * hs * ws
    base = torch.stack(torch.meshgrid(torch.arange(ws).float(), torch.arange(hs).float(), torch.arange(5).float(),
                                      indexing="xy"), 0).permute(0, 3, 1, 2)
    xyz = (base.unsqueeze(0).repeat(G, 1, 1, 1, 1) * 0.02 + 1.0 + 0.004 * torch.randn(G, 3, 5, hs, ws)).to(dev)
    _, codes = knn_lattice(xyz, 5, 16, with_codes=True, with_idx=False)
    for C in (32, 64):
        LE = torch.randn(G * Ng, 2 * C, device=dev)
        T = pointflow.stat_blocks(G, Ng)
        part = torch.empty((G, T, C, 2), dtype=torch.float64, device=dev)
        scale = torch.rand(G, 2 * C, device=dev) + 0.5
        shift = torch.randn(G, 2 * C, device=dev) * 0.1
        Y = torch.empty((G * Ng, 224), device=dev)

        def stats():
            _lib.call("pf_edge_stats_f32", _lib.ptr(LE), 2 * C, C, None, 16, G, Ng, _lib.ptr(part), _lib.ptr(codes), 5, hs, ws,
                      _lib.stream())

        def apply():
            _lib.call("pf_edge_apply_f32", _lib.ptr(LE), 2 * C, C, None, 16, G, Ng, _lib.ptr(scale), _lib.ptr(shift), 2 * C, 1, 1,
                      _lib.ptr(Y), 224, _lib.ptr(codes), 5, hs, ws, _lib.stream())

        ts, ta = timeit(stats), timeit(apply)
        gb = G * Ng * 16 * C * 4 / 1e9
        print("G=%d %dx%d C=%d: stats %.1f us (%.1f TB/s of neighbour rows), apply %.1f us (%.1f TB/s), checksum %.9e %.9e"
              % (G, hs, ws, C, ts, gb / ts * 1e3, ta, gb / ta * 1e3, float(part.sum()), float(Y[:, :2 * C].double().sum())),
              flush=True)
