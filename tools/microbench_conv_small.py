"""Micro-benchmark of pf_conv2d_small_f32 on the conv0.1 shape (3 views, 8->8, 512x640) for counter runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointmvsnet_amd import pointflow
dev = torch.device("cuda:0")
x = torch.randn(3, 8, 512, 640, device=dev)
conv = torch.nn.Conv2d(8, 8, 3, padding=1, bias=False).to(dev)
sc = torch.rand(3, 8, device=dev) + 0.5
sh = torch.randn(3, 8, device=dev) * 0.1
for _ in range(3):
    pointflow.conv2d_small(x, conv, (sc, sh), 1, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    pointflow.conv2d_small(x, conv, (sc, sh), 1, True)
e1.record()
torch.cuda.synchronize()
print("conv2d_small 8->8 512x640x3: %.1f us per call" % (e0.elapsed_time(e1) * 1000 / 20))
