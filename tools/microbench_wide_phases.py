"""Phase ablation of the tower kernels (csrc/conv2d_wide.hip built with -DPF_DBG_NOLOAD / NOSTORE / NOMFMA: library
variants under pointmvsnet_amd/build/variants, see tools/jobs/gpurun_job_phases.sh): time per launch on the cfg2
shapes, 6 samples (two towers' worth), with the previous BatchNorm applied from affine rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import pointflow  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=60):
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


LAYERS = [("conv0.0x2", 3, 16, 512, 640, 3, 1, 3), ("conv0.1", 8, 8, 512, 640, 3, 1, 6), ("conv1.0", 8, 16, 512, 640, 5, 2, 6),
          ("conv1.1", 16, 16, 256, 320, 3, 1, 6), ("conv2.0", 16, 32, 256, 320, 5, 2, 6), ("conv2.1", 32, 32, 128, 160, 3, 1, 6),
          ("conv3.0", 32, 64, 128, 160, 5, 2, 6), ("conv3.1", 64, 64, 64, 80, 3, 1, 6)]
out = []
for name, cin, cout, h, w, ks, stride, views in LAYERS:
    conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
    x = torch.randn(views, cin, h, w, device=dev)
    aff = None if cin == 3 else (torch.rand(views, cin, device=dev) + 0.5, torch.randn(views, cin, device=dev) * 0.1)
    t = timeit(lambda: pointflow.conv2d_wide(x, conv, aff, 1, True))
    out.append("%s %.1f" % (name, t))
print(" | ".join(out), flush=True)
