"""Micro-benchmark of csrc/conv2d_wide.hip on the cfg2 tower shapes (3 views batched, and 6 = two towers' worth)
beside the library convolution (which has no fused BatchNorm statistics and needs the previous BatchNorm+ReLU applied
by a separate pass)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pointmvsnet_amd import pointflow  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


LAYERS = [("conv0.0", 3, 8, 512, 640, 3, 1), ("conv0.1", 8, 8, 512, 640, 3, 1), ("conv1.0", 8, 16, 512, 640, 5, 2), ("conv1.1", 16, 16, 256, 320, 3, 1), ("conv2.0", 16, 32, 256, 320, 5, 2), ("conv2.1", 32, 32, 128, 160, 3, 1),
          ("conv3.0", 32, 64, 128, 160, 5, 2), ("conv3.1", 64, 64, 64, 80, 3, 1)]
for views in (3, 6):
    for name, cin, cout, h, w, ks, stride in LAYERS:
        conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
        x = torch.randn(views, cin, h, w, device=dev)
        sc = torch.rand(views, cin, device=dev) + 0.5
        sh = torch.randn(views, cin, device=dev) * 0.1
        xin = F.relu(x * sc.view(views, cin, 1, 1) + sh.view(views, cin, 1, 1))
        ref = F.conv2d(xin.double(), conv.weight.double(), None, stride, ks // 2)
        aff = None if cin == 3 else (sc, sh)
        if cin == 3:
            xin = x
            ref = F.conv2d(xin.double(), conv.weight.double(), None, stride, ks // 2)
        y, _ = pointflow.conv2d_wide(x, conv, aff, 1, True)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        flops = 2.0 * ref.numel() * ks * ks * cin
        tw = timeit(lambda: pointflow.conv2d_wide(x, conv, aff, 1, True))
        tl = timeit(lambda: conv(xin))
        print("%d views %s %d->%d %dx%d k%d s%d: wide %.1f us (%.1f TF, rel err %.1e) | library %.1f"
              % (views, name, cin, cout, h, w, ks, stride, tw, flops / tw * 1e-6, err, tl), flush=True)
