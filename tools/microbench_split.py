"""The bf16x3 tower kernel (csrc/conv2d_wide.hip, conv2d_wide_split_kernel; PF_MATRIX_SPLIT) beside the exact-f32 kernel
on the cfg2 tower shapes (3 views, and 6 = both towers' samples): stand-alone microseconds, TFLOP/s, and the largest error
of each against a float64 convolution.  VERDICT r4 item 4's measurement; built in round 5, NOT yet run on hardware
(the round lost its GPU access -- DESIGN.md section 7): run it with

    gpurun -- 'python tools/microbench_split.py > gpurun_out/microbench_split.log 2>&1'

The gate for adoption is written in DESIGN.md: >= 1.4 x on the kernel, teacher-forced max norm < 1e-5 with it switched on
(PF_MATRIX_SPLIT=1 python -m pytest tests/test_gpu_teacher.py).
"""
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


LAYERS = [("conv2.0", 16, 32, 256, 320, 5, 2), ("conv2.1", 32, 32, 128, 160, 3, 1), ("conv3.0", 32, 64, 128, 160, 5, 2),
          ("conv3.1", 64, 64, 64, 80, 3, 1)]
for views in (3, 6):
    for name, cin, cout, h, w, ks, stride in LAYERS:
        conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
        x = torch.randn(views, cin, h, w, device=dev)
        sc = torch.rand(views, cin, device=dev) + 0.5
        sh = torch.randn(views, cin, device=dev) * 0.1
        xin = F.relu(x * sc.view(views, cin, 1, 1) + sh.view(views, cin, 1, 1))
        ref = F.conv2d(xin.double(), conv.weight.double(), None, stride, ks // 2)
        flops = 2.0 * ref.numel() * ks * ks * cin
        out = {}
        for split in (0, 1):
            pointflow.MATRIX_SPLIT = split
            y, part = pointflow.conv2d_wide(x, conv, (sc, sh), 1, True)
            err = float((y.double() - ref).abs().max() / ref.abs().max())
            stat = float((part.sum(dim=1)[..., 0] - ref.sum(dim=(2, 3))).abs().max() / ref.sum(dim=(2, 3)).abs().max())
            t = timeit(lambda: pointflow.conv2d_wide(x, conv, (sc, sh), 1, True))
            out[split] = (t, err, stat)
        pointflow.MATRIX_SPLIT = 0
        print("%d views %s %d->%d %dx%d k%d s%d: f32 %.1f us (%.1f TF, err %.1e) | bf16x3 %.1f us (%.1f TF of useful flop, "
              "err %.1e, stats %.1e) | x%.2f"
              % (views, name, cin, cout, h, w, ks, stride, out[0][0], flops / out[0][0] * 1e-6, out[0][1], out[1][0],
                 flops / out[1][0] * 1e-6, out[1][1], out[1][2], out[0][0] / out[1][0]), flush=True)
