"""Tower kernels (csrc/conv2d_wide.hip) at the cfg2 shapes, both towers' worth of views (6), stand-alone: us per launch.
Run once per ablation library (tools/experiments/build_dbg_variants.sh; PF_LIB_PATH selects it) to see what a layer's
time is made of: global loads of the patch, the matrix instructions, the output stores."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import pointflow  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=50):
    for _ in range(10):
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
views = 6
for name, cin, cout, h, w, ks, stride in LAYERS:
    conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
    x = torch.randn(views, cin, h, w, device=dev)
    sc = torch.rand(views, cin, device=dev) + 0.5
    sh = torch.randn(views, cin, device=dev) * 0.1
    aff = None if cin == 3 else (sc, sh)
    tw = timeit(lambda: pointflow.conv2d_wide(x, conv, aff, 1, True))
    flops = 2.0 * views * cout * (h // stride) * (w // stride) * ks * ks * cin
    print("tower %s %d->%d k%d s%d: %.1f us  %.1f TF" % (name, cin, cout, ks, stride, tw, flops / tw * 1e-6), flush=True)
