"""Workload for SQ-counter passes on single kernels: runs a few launches of selected entry points on the cfg2
shapes (tower convs, the chain's GEMMs) so that `rocprofv3 --pmc ...` attributes counters to them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pointmvsnet_amd import pointflow  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
which = sys.argv[1:] or ["wide"]
if "wide" in which:
    for cin, cout, h, w, ks, stride in ((16, 16, 256, 320, 3, 1), (32, 32, 128, 160, 3, 1), (32, 64, 128, 160, 5, 2),
                                        (64, 64, 64, 80, 3, 1)):
        conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
        x = torch.randn(3, cin, h, w, device=dev)
        sc = torch.rand(3, cin, device=dev) + 0.5
        sh = torch.randn(3, cin, device=dev) * 0.1
        for _ in range(4):
            pointflow.conv2d_wide(x, conv, (sc, sh), 1, True)
if "lowch" in which:      # the towers' full-resolution / 16-channel layers, both towers' worth of samples
    for cin, cout, h, w, ks, stride, views in ((3, 16, 512, 640, 3, 1, 3), (8, 8, 512, 640, 3, 1, 6), (8, 16, 512, 640, 5, 2, 6),
                                               (16, 16, 256, 320, 3, 1, 6)):
        conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
        x = torch.randn(views, cin, h, w, device=dev)
        aff = None if cin == 3 else (torch.rand(views, cin, device=dev) + 0.5, torch.randn(views, cin, device=dev) * 0.1)
        for _ in range(4):
            pointflow.conv2d_wide(x, conv, aff, 1, True)
if "gemm" in which:
    for K, Nc, ldx in ((136, 64, 136), (224, 64, 224), (64, 128, 224)):
        X = torch.randn(4 * 25600, ldx, device=dev)
        Wt = torch.randn(K, Nc, device=dev) * 0.1
        Y = torch.empty(4 * 25600, Nc, device=dev)
        for _ in range(4):
            pointflow.pointwise_gemm(X, True, ldx, Wt, Y, Nc, 4, 25600, K, Nc, want_stats=True)
if "conv3d" in which:
    w = torch.randn(8, 64, 3, 3, 3, device=dev) * 0.05
    x = torch.randn(1, 64, 48, 64, 80, device=dev)
    for _ in range(4):
        pointflow.conv3d_k3(x, w, 1, True)
if "volume" in which:     # VolumeConv's forward (the eleven launches of forward_fused) at config 2's cost volume
    from pointmvsnet_amd import synthetic
    from pointmvsnet_amd.model import PointMVSNet
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    vc = net.coarse_vol_conv.to(dev).train()
    cost = torch.randn(1, 64, 48, 64, 80, device=dev)
    with torch.no_grad():
        for _ in range(3):
            vc.forward_fused(cost)
if "wgrad" in which:      # the training step's weight gradients at BASELINE config 4's shapes
    from pointmvsnet_amd import train_ops
    for N, co, ci, sp, k, st in ((1, 8, 64, (48, 64, 80), 3, 1), (3, 8, 8, (512, 640), 3, 1), (3, 16, 16, (256, 320), 3, 1),
                                 (3, 64, 64, (64, 80), 3, 1), (1, 16, 64, (48, 64, 80), 3, 2)):
        nd = len(sp)
        x = torch.randn(N, ci, *sp, device=dev)
        dy = torch.randn(N, co, *[(v - 1) // st + 1 for v in sp], device=dev)
        for _ in range(3):
            train_ops.conv_wgrad(dy, x, (k,) * nd, st, (k // 2,) * nd)
    g, X = torch.randn(102400, 128, device=dev), torch.randn(102400, 136, device=dev)
    for _ in range(3):
        train_ops.rows_wgrad(g, X, 128, 136)
torch.cuda.synchronize()
