"""Micro-benchmark of pf_knn_lattice_f32 on the two flow lattices of cfg2 (G x 5 x 64 x 80 points, window 5, k 16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd.utils.torch_utils import knn_lattice  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=20):
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


for G in (1, 4):
    # a tilted plane + noise: realistic spacing (lattice neighbours are spatial neighbours) with no exact ties
    zz, yy, xx = torch.meshgrid(torch.arange(5.0), torch.arange(64.0), torch.arange(80.0), indexing="ij")
    base = torch.stack([xx * 1.7, yy * 1.7, 600 + zz * 2.1 + 0.01 * xx], 0)
    xyz = (base.unsqueeze(0) + 0.3 * torch.randn(G, 3, 5, 64, 80)).to(dev).contiguous()
    print("G=%d: sorting-network kNN (codes + int64 indices) %.1f us, codes only %.1f us"
          % (G, timeit(lambda: knn_lattice(xyz, 5, 16, with_codes=True)),
             timeit(lambda: knn_lattice(xyz, 5, 16, with_codes=True, with_idx=False))), flush=True)
