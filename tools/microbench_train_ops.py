"""Stand-alone timings of the training step's backward kernels at BASELINE config 4's shapes (one scene, 3 views,
640x512, 48 depth planes): every weight gradient, the data gradients, the BatchNorm backward, the warp backward.

    python tools/microbench_train_ops.py [--reps 20] > gpurun_out/microbench_train_ops.log

Prints one line per call: microseconds (HIP events around `reps` back-to-back launches) and TFLOP/s where the call is
a contraction (against the 157.3 TF f32 peak of MI355X).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointmvsnet_amd import train_ops  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    r = lambda *s: torch.randn(*s, device=dev)   # noqa: E731
    V, H, W, D = 3, 512, 640, 48
    total = 0.0
    print("# weight gradients: name, us, TFLOP/s")
    cases = [("tower 3->8 k3", V, 8, 3, (H, W), 3, 1), ("tower 8->8 k3", V, 8, 8, (H, W), 3, 1),
             ("tower 8->16 k5s2", V, 16, 8, (H, W), 5, 2), ("tower 16->16 k3", V, 16, 16, (H // 2, W // 2), 3, 1),
             ("tower 16->32 k5s2", V, 32, 16, (H // 2, W // 2), 5, 2), ("tower 32->32 k3", V, 32, 32, (H // 4, W // 4), 3, 1),
             ("tower 32->64 k5s2", V, 64, 32, (H // 4, W // 4), 5, 2), ("tower 64->64 k3", V, 64, 64, (H // 8, W // 8), 3, 1),
             ("vol conv0_1 64->8", 1, 8, 64, (D, 64, 80), 3, 1), ("vol conv1_0 64->16 s2", 1, 16, 64, (D, 64, 80), 3, 2),
             ("vol conv2_0 16->32 s2", 1, 32, 16, (24, 32, 40), 3, 2), ("vol conv3_0 32->64 s2", 1, 64, 32, (12, 16, 20), 3, 2),
             ("vol conv3_1 64->64", 1, 64, 64, (6, 8, 10), 3, 1), ("vol conv1_1 16->16", 1, 16, 16, (24, 32, 40), 3, 1),
             ("vol conv2_1 32->32", 1, 32, 32, (12, 16, 20), 3, 1), ("vol conv6_2 8->1", 1, 1, 8, (D, 64, 80), 3, 1)]
    for name, N, Co, Ci, sp, k, s in cases:
        nd = len(sp)
        x = r(N, Ci, *sp)
        osp = tuple((v - 1) // s + 1 for v in sp)
        dy = r(N, Co, *osp)
        us = timed(lambda: train_ops.conv_wgrad(dy, x, (k,) * nd, s, (k // 2,) * nd), a.reps)
        fl = 2.0 * N * Co * Ci * (k ** nd)
        for v in osp:
            fl *= v
        times = {"tower": 2}.get(name.split()[0], 1)          # both towers run it
        total += us * times
        print("wgrad %-26s %9.1f us  %6.1f TF/s  (x%d per step)" % (name, us, fl / us / 1e6, times))
    for name, Ci, Co, sp in [("vol deconv4_0 64->32", 64, 32, (6, 8, 10)), ("vol deconv5_0 32->16", 32, 16, (12, 16, 20)),
                             ("vol deconv6_0 16->8", 16, 8, (24, 32, 40))]:
        x = r(1, Ci, *sp)
        dy = r(1, Co, *(2 * v for v in sp))
        us = timed(lambda: train_ops.conv_wgrad(x, dy, (3, 3, 3), 2, (1, 1, 1)), a.reps)
        fl = 2.0 * 27 * Ci * Co * sp[0] * sp[1] * sp[2]
        total += us
        print("wgrad %-26s %9.1f us  %6.1f TF/s" % (name, us, fl / us / 1e6))
    for name, P, Cg, Cx in [("E0 64x136 @102400", 102400, 64, 136), ("E1 64x32", 102400, 64, 32), ("E2 128x64", 102400, 128, 64),
                            ("mlp1 64x224", 102400, 64, 224), ("mlp2 64x64", 102400, 64, 64), ("mlp3 16x64", 102400, 16, 64)]:
        g, x = r(P, Cg), r(P, Cx)
        us = timed(lambda: train_ops.rows_wgrad(g, x, Cg, Cx), a.reps)
        total += us * 1.25                                   # + the 25 600-point iteration
        print("wgrad rows %-21s %9.1f us  %6.1f TF/s" % (name, us, 2.0 * P * Cg * Cx / us / 1e6))
    print("# weight gradients per step (sum): %.0f us" % total)
    print("# data gradients")
    w = lambda *s: torch.randn(*s, device=dev) * 0.1   # noqa: E731
    for name, N, Co, Ci, sp, k, s in cases[1:8]:
        osp = tuple((v - 1) // s + 1 for v in sp)
        dy, wt = r(N, Co, *osp), w(Co, Ci, k, k)
        us = timed(lambda: train_ops.conv2d_dgrad(dy, wt, s), a.reps)
        fl = 2.0 * N * Co * Ci * k * k * osp[0] * osp[1]
        print("dgrad %-26s %9.1f us  %6.1f TF/s (incl. weight flip / pack)" % (name, us, fl / us / 1e6))
    dy, wt = r(1, 8, D, 64, 80), w(8, 64, 3, 3, 3)
    us = timed(lambda: train_ops.conv3d_dgrad_flip(dy, wt), a.reps)
    print("dgrad %-26s %9.1f us  %6.1f TF/s" % ("vol conv0_1 (8->64, 2 x 8->32)", us,
                                                 2.0 * 27 * 64 * 8 * D * 64 * 80 / us / 1e6))
    from pointmvsnet_amd import pointflow
    dy, wt = r(1, 16, 24, 32, 40), w(16, 64, 3, 3, 3)
    us = timed(lambda: pointflow.deconv3d_k3s2(dy, None, wt, False), a.reps)
    print("dgrad %-26s %9.1f us  %6.1f TF/s" % ("vol conv1_0 (16->64 deconv)", us, 2.0 * 27 * 64 * 16 * 24 * 32 * 40 / us / 1e6))
    print("# BatchNorm backward (reduce + coeffs + apply), planar")
    for name, N, C, sp in [("tower 8ch full", V, 8, (H, W)), ("tower 16ch half", V, 16, (H // 2, W // 2)),
                           ("tower 64ch eighth", V, 64, (H // 8, W // 8)), ("vol 8ch full", 1, 8, (D, 64, 80))]:
        y, g = r(N, C, *sp), r(N, C, *sp)
        rows = torch.rand(4, N, C, device=dev) + 0.5
        us = timed(lambda: train_ops.bn_backward(g, y, rows, 1, True), a.reps)
        nbytes = 20.0 * y.numel()
        print("bn_bwd %-25s %9.1f us  %6.2f TB/s" % (name, us, nbytes / us / 1e6))
    y, g = r(102400, 64), r(102400, 64)
    rows = torch.rand(4, 1, 64, device=dev) + 0.5
    us = timed(lambda: train_ops.rows_bn_backward(g, y, rows, 64, 1, 102400, 1, True), a.reps)
    print("bn_bwd rows 102400x64            %9.1f us  %6.2f TB/s" % (us, 20.0 * y.numel() / us / 1e6))


if __name__ == "__main__":
    main()
