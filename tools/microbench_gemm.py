"""Micro-benchmark of pf_pointwise_gemm_f32 on the six GEMM shapes of one flow iteration (cfg2: flow-1 G=1,
flow-2 G=4, Ng=25600): the chunked-through-LDS kernel (channel-major input) against the direct-A kernel (point-major rows)."""
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


SHAPES = [("E0", 136, 64, 136, False), ("E1", 32, 64, 224, False), ("E2", 64, 128, 224, False),
          ("M1", 224, 64, 224, True), ("M2", 64, 64, 64, True), ("M3", 64, 32, 64, True)]
for G in (1, 4):
    Ng = 25600
    for name, K, Nc, ldx, affine in SHAPES:
        X = torch.randn(G * Ng, ldx, device=dev)
        Wt = torch.randn(K, Nc, device=dev) * 0.1
        Y = torch.empty(G * Ng, Nc, device=dev)
        aff = (torch.rand(G, K, device=dev) + 0.5, torch.randn(G, K, device=dev) * 0.1) if affine else None
        ref = None
        line = "G=%d %s K=%d Nc=%d:" % (G, name, K, Nc)
        flops = 2.0 * G * Ng * K * Nc
        X_cm = X[:, :K].reshape(G, Ng, K).transpose(1, 2).contiguous()
        for pm in (False, True):                           # chunked-through-LDS kernel (channel-major input), then direct-A
            args = (X if pm else X_cm, pm, ldx if pm else 0, Wt, Y, Nc, G, Ng, K, Nc)
            pointflow.pointwise_gemm(*args, in_affine=aff, want_stats=True)
            if ref is None:
                ref = Y.clone()
            err = float((Y - ref).abs().max()) / float(ref.abs().max())
            t = timeit(lambda: pointflow.pointwise_gemm(*args, in_affine=aff, want_stats=True))
            line += "  [%s] %.1f us %.1f TF (rel diff %.1e)" % ("direct" if pm else "chunked", t, flops / t / 1e6, err)
        print(line, flush=True)
