"""Planning experiment: what would S scenes per launch buy the tower and VolumeConv kernels?  Times the eleven tower
layers at 6 / 12 / 24 samples (one, two, four scenes' worth of both towers) and VolumeConv's fused forward at batch
1 / 2 / 4 on the cfg2 shapes, per scene.  (The kernels already take samples_per_stat, i.e. per-scene statistics.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import pointflow, synthetic  # noqa: E402
from pointmvsnet_amd.networks import VolumeConv  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


LAYERS = [("conv0.1", 8, 8, 512, 640, 3, 1), ("conv1.0", 8, 16, 512, 640, 5, 2), ("conv1.1", 16, 16, 256, 320, 3, 1),
          ("conv1.2", 16, 16, 256, 320, 3, 1), ("conv2.0", 16, 32, 256, 320, 5, 2), ("conv2.1", 32, 32, 128, 160, 3, 1),
          ("conv2.2", 32, 32, 128, 160, 3, 1), ("conv3.0", 32, 64, 128, 160, 5, 2), ("conv3.1", 64, 64, 64, 80, 3, 1),
          ("conv3.2", 64, 64, 64, 80, 3, 1)]
for samples in (6, 12, 24):
    total = 0.0
    for name, cin, cout, h, w, ks, stride in LAYERS:
        conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
        x = torch.randn(samples, cin, h, w, device=dev)
        aff = (torch.rand(samples, cin, device=dev) + 0.5, torch.randn(samples, cin, device=dev) * 0.1)
        total += timeit(lambda: pointflow.conv2d_wide(x, conv, aff, 1, True))
    print("towers without the first layer, %2d samples per launch: %7.1f us per launch set = %6.1f us per scene"
          % (samples, total, total * 6 / samples), flush=True)
vc = VolumeConv(64, 8)
synthetic.seed_weights(vc, 0)
vc = vc.to(dev).train()
for B in (1, 2, 4):
    x = torch.randn(B, 64, 48, 64, 80, device=dev)
    with torch.no_grad():
        t = timeit(lambda: (vc.forward_fused(x), pointflow.flush_counters()), reps=20)
    print("VolumeConv.forward_fused, batch %d: %7.1f us = %6.1f us per scene" % (B, t, t / B), flush=True)
