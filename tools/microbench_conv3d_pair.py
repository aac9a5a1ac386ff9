"""conv0_1 of VolumeConv (64 -> 8 on 48x64x80, cfg2; 64 -> 8 on 96x120x160, cfg3): the paired-rows kernel
(csrc/conv3d_pair.hip) against the 16-channel-wide tile of csrc/conv3d.hip."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import pointflow  # noqa: E402

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


for D, H, W in ((48, 64, 80), (96, 120, 160)):
    x = torch.randn(1, 64, D, H, W, device=dev)
    w = torch.randn(8, 64, 3, 3, 3, device=dev) * 0.02
    flops = 2.0 * D * H * W * 27 * 64 * 8
    ref = torch.nn.functional.conv3d(x.double(), w.double(), None, 1, 1)
    y, _ = pointflow.conv3d_k3(x, w, 1, True)                      # (64 -> 8: the paired-rows kernel, conv3d_pair.hip)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: pointflow.conv3d_k3(x, w, 1, True))
    print("64->8 on %dx%dx%d: %.1f us %.1f TF (rel err vs float64 %.1e)" % (D, H, W, t, flops / t / 1e6, err), flush=True)
