"""Micro-benchmark of the feature-tower convolutions on the cfg2 shapes (3 views batched): the f32-MFMA kernel
(pf_conv2d_f32, every tile variant via PF_CONV2D_VARIANT = 100*TR + 10*KG + MINW), the direct-FMA kernel for
8/16 output channels (pf_conv2d_small_f32) and the library convolution (no fused BatchNorm statistics)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

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


LAYERS = [("conv0.0", 3, 8, 512, 640, 3, 1), ("conv0.1", 8, 8, 512, 640, 3, 1), ("conv1.0", 8, 16, 512, 640, 5, 2),
          ("conv1.1", 16, 16, 256, 320, 3, 1), ("conv2.0", 16, 32, 256, 320, 5, 2), ("conv2.1", 32, 32, 128, 160, 3, 1),
          ("conv3.0", 32, 64, 128, 160, 5, 2), ("conv3.1", 64, 64, 64, 80, 3, 1)]
for name, cin, cout, h, w, ks, stride in LAYERS:
    conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
    x = torch.randn(3, cin, h, w, device=dev)
    sc = torch.rand(3, cin, device=dev) + 0.5
    sh = torch.randn(3, cin, device=dev) * 0.1
    xin = F.relu(x * sc.view(3, cin, 1, 1) + sh.view(3, cin, 1, 1))
    ref = F.conv2d(xin[:, :, :40, :56].double(), conv.weight.double(), None, stride, ks // 2)
    line = "%s %d->%d %dx%d k%d s%d:" % (name, cin, cout, h, w, ks, stride)
    variants = [0, 1536, 768, 512, 384, 256] if cout <= 32 else [0]
    for v in variants:
        if v:
            os.environ["PF_CONV2D_CAP"] = str(v)
        else:
            os.environ.pop("PF_CONV2D_CAP", None)
        y, _ = pointflow.conv2d(x[:, :, :40, :56].contiguous(), conv, (sc, sh), 1, True)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: pointflow.conv2d(x, conv, (sc, sh), 1, True))
        line += "  mfma[%d] %.1f%s" % (v, t, "" if err < 1e-5 else " (ERR %.1e)" % err)
    os.environ.pop("PF_CONV2D_CAP", None)
    if pointflow.conv2d_small_preferred(conv):
        line += "  | small %.1f" % timeit(lambda: pointflow.conv2d_small(x, conv, (sc, sh), 1, True))
    line += "  | library %.1f us" % timeit(lambda: conv(xin))
    print(line, flush=True)
