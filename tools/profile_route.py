"""Where the DROP-IN route's time goes: the reference's unmodified model.py (oracle/_ref/reference_model_py.txt) on this
package's operator layer (pointmvsnet_amd.compat), BASELINE cfg 2, eager.

    python tools/profile_route.py [--maps 12] [--out gpurun_out/route_profile.md]

Three views of the same loop of whole forwards:
  1. wall time per depth map (device-synchronised), and the same with the host timeline only (time until the last launch
     is ENQUEUED): when the two agree the route is bound by the host's launch rate, not by the kernels;
  2. cProfile of the loop, top functions by own time and by cumulative time (the Python side: ctypes calls of this package,
     ATen calls of model.py, torch.inverse / linspace / .to() synchronisations);
  3. torch.autograd.profiler's per-operator host / device totals (which ATen operators of model.py dominate).
A rocprofv3 --kernel-trace --stats run of this script (PROFILE=0: views 2 and 3 off) gives the kernel side.
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pointmvsnet_amd import _lib, compat, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maps", type=int, default=12)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--model-py", default=os.path.join(ROOT, "oracle", "_ref", "reference_model_py.txt"))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _lib.load()
    net = compat.load_reference_model(args.model_py).PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    scenes = []
    for i in range(4):
        data, img_scales, inter_scales = synthetic.make_config(args.config, seed=i)
        scenes.append({k: v.to(dev) for k, v in data.items()})
    devnull = open(os.devnull, "w")

    def forward(i):
        saved = sys.stdout
        sys.stdout = devnull                      # model.py prints "flow: i" per iteration
        try:
            with torch.no_grad():
                return net(scenes[i % 4], img_scales, inter_scales, isFlow=True, isTest=True)
        finally:
            sys.stdout = saved

    for i in range(3):
        forward(i)
    torch.cuda.synchronize()
    lines = []
    t0 = time.perf_counter()
    for i in range(args.maps):
        forward(i)
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    lines.append("# drop-in route (reference model.py on the HIP operator layer), %s, eager\n" % args.config)
    lines.append("%d depth maps: %.2f ms per map device-synchronised = %.1f maps/s; host enqueue time %.2f ms per map "
                 "(%.0f %% of the wall time)\n" % (args.maps, wall / args.maps * 1e3, args.maps / wall,
                                                   issued / args.maps * 1e3, 100.0 * issued / wall))
    if os.environ.get("PROFILE", "1") == "1":
        prof = cProfile.Profile()
        prof.enable()
        for i in range(args.maps):
            forward(i)
        torch.cuda.synchronize()
        prof.disable()
        for key, title in (("tottime", "own time"), ("cumulative", "cumulative time")):
            buf = io.StringIO()
            pstats.Stats(prof, stream=buf).strip_dirs().sort_stats(key).print_stats(45)
            lines.append("\n## cProfile, %d maps, by %s\n\n```\n%s\n```\n" % (args.maps, title, buf.getvalue()[-9000:]))
        try:
            from torch.autograd import profiler
            with profiler.profile(use_device="cuda") as p:
                for i in range(4):
                    forward(i)
                torch.cuda.synchronize()
            lines.append("\n## torch profiler, 4 maps, operators by host time\n\n```\n%s\n```\n"
                         % p.key_averages().table(sort_by="self_cpu_time_total", row_limit=40)[-14000:])
        except Exception as exc:
            lines.append("\n(torch profiler unavailable: %r)\n" % (exc,))
    text = "".join(lines)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    print(text[:6000])


if __name__ == "__main__":
    main()
