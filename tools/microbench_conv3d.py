"""Micro-benchmark of pf_conv3d_k3_f32 on VolumeConv's two big layers (cfg2: 64 ch, 48x64x80 voxels) for every
tile-depth / occupancy variant (PF_CONV3D_VARIANT = 10*TD + MINW, read by the library at each call), with a
float64 check of each variant on a sub-volume."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pointmvsnet_amd import pointflow  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(1, 64, 48, 64, 80, device=dev)
xs = x[:, :, :9, :13, :37].contiguous()


def run(cout, stride, variants):
    w = torch.randn(cout, 64, 3, 3, 3, device=dev) * 0.05
    ref = F.conv3d(xs.double(), w.double(), stride=stride, padding=1)
    for v in variants:
        if v:
            os.environ["PF_CONV3D_VARIANT"] = str(v)
        else:
            os.environ.pop("PF_CONV3D_VARIANT", None)
        y, part = pointflow.conv3d_k3(xs, w, stride, True)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        s_err = float((part.sum(1)[0, :, 0] - ref.sum((0, 2, 3, 4))).abs().max())
        for _ in range(3):
            pointflow.conv3d_k3(x, w, stride, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            pointflow.conv3d_k3(x, w, stride, True)
        e1.record()
        torch.cuda.synchronize()
        print("conv3d 64->%d stride %d variant %2d: %7.1f us  rel err %.1e  stat err %.1e"
              % (cout, stride, v, e0.elapsed_time(e1) * 1000 / 20, err, s_err), flush=True)


run(8, 1, [0, 24])
run(16, 2, [0, 12])
os.environ.pop("PF_CONV3D_VARIANT", None)

# the inner VolumeConv layers (small volumes): ours against the library convolution
import torch.nn as nn  # noqa: E402


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


for name, cin, cout, d, h, w, stride in (("conv1_1", 16, 16, 24, 32, 40, 1), ("conv2_0", 16, 32, 24, 32, 40, 2),
                                         ("conv2_1", 32, 32, 12, 16, 20, 1)):
    conv = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False).to(dev)
    xx = torch.randn(1, cin, d, h, w, device=dev)
    ref = conv(xx).double()
    y, _ = pointflow.conv3d_k3(xx, conv.weight, stride, True)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print("%s %d->%d on %dx%dx%d stride %d: ours %.1f us (rel diff to library %.1e) | library %.1f us"
          % (name, cin, cout, d, h, w, stride, timeit(lambda: pointflow.conv3d_k3(xx, conv.weight, stride, True)),
             err, timeit(lambda: conv(xx))), flush=True)
