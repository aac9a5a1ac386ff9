"""Micro-benchmark of the VolumeConv decoder kernels on the cfg2 shapes: pf_deconv3d_k3s2_f32 (conv5_0 / conv6_0,
with the skip add) and pf_conv3d_k3_few_f32 (conv6_2, without the skip add: adding on load doubles its tap loads, 63 us vs 25 us)."""
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


for name, cin, cout, d, h, w in (("conv4_0", 64, 32, 6, 8, 10), ("conv5_0", 32, 16, 12, 16, 20),
                                 ("conv6_0", 16, 8, 24, 32, 40), ("dgrad conv1_0", 16, 64, 24, 32, 40),
                                 ("dgrad conv2_0", 32, 16, 12, 16, 20)):
    xa = torch.randn(1, cin, d, h, w, device=dev)
    xb = torch.randn(1, cin, d, h, w, device=dev)
    wt = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
    deconv = torch.nn.ConvTranspose3d(cin, cout, 3, 2, 1, 1, bias=False).to(dev)
    print("%s %d->%d on %dx%dx%d: hip+skip+stats %.1f us | hip %.1f us | library add+deconv %.1f us"
          % (name, cin, cout, d, h, w, timeit(lambda: pointflow.deconv3d_k3s2(xa, xb, wt, True)),
             timeit(lambda: pointflow.deconv3d_k3s2(xa, None, wt, False)), timeit(lambda: deconv(xa + xb))), flush=True)

x = torch.randn(1, 8, 48, 64, 80, device=dev)
wf = torch.randn(1, 8, 3, 3, 3, device=dev) * 0.1
print("conv6_2 8->1 on 48x64x80: few %.1f us"
      % timeit(lambda: pointflow.conv3d_k3_few(x, wf)))
