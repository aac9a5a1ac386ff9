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
    line = "64->8 on %dx%dx%d:" % (D, H, W)
    ref = None
    for pair, minw in ((0, 0), (1, 4), (1, 3), (1, 2)):
        pointflow.CONV3D_PAIR = pair
        if minw:
            os.environ["PF_CONV3D_PAIR_MINW"] = str(minw)
        y, _ = pointflow.conv3d_k3(x, w, 1, True)
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: pointflow.conv3d_k3(x, w, 1, True))
        line += "  [%s] %.1f us %.1f TF (diff %.1e)" % ("wide16" if not pair else "pair/minw%d" % minw, t, flops / t / 1e6, err)
    print(line, flush=True)
