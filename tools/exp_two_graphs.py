"""Experiment: do two captured forwards replayed on two streams overlap on MI355X?  (scene-level concurrency)

    python tools/exp_two_graphs.py [--config cfg2] [--steps 200]

Prints depth maps/s for (a) one graph on one stream, (b) two graphs alternating on ONE stream, (c) two graphs on two
streams.  Each graph owns its static input, its plan block and its intermediates; the model (weights, BatchNorm
buffers) is shared -- the running statistics then see the scenes in a non-deterministic order, which is why this is
an experiment and not the product path."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointmvsnet_amd import synthetic  # noqa: E402
from pointmvsnet_amd.graph import GraphedForward  # noqa: E402
from pointmvsnet_amd.model import PointMVSNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--onegraph", type=int, default=0, help="also capture this many scenes as branches of ONE graph")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    _, _, _, _, _, img_scales, inter_scales = synthetic.CONFIGS[args.config]
    scenes = []
    for seed in range(4):
        data, _, _ = synthetic.make_config(args.config, seed=seed)
        b = {k: v.to(dev) for k, v in data.items()}
        b["cam_params_list_host"] = data["cam_params_list"]
        b["mean_host"], b["std_host"] = data["mean"], data["std"]
        scenes.append(b)
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    streams = [torch.cuda.Stream() for _ in range(args.lanes)]
    graphs = []
    with torch.no_grad():
        for lane in range(args.lanes):
            with torch.cuda.stream(streams[lane]):
                graphs.append(GraphedForward(net, scenes[lane], img_scales, inter_scales, isFlow=True, isTest=True))
    torch.cuda.synchronize()

    def run(mode):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(args.steps):
                lane = 0 if mode == "one" else i % args.lanes
                st = streams[0] if mode in ("one", "alternate") else streams[lane]
                with torch.cuda.stream(st):
                    graphs[lane](scenes[i % 4])
        torch.cuda.synchronize()
        return args.steps / (time.perf_counter() - t0)

    # host cost of one step's pieces (nothing waits for the GPU in here except the pinned-block reuse)
    acc = {"fill": 0.0, "upload+copy": 0.0, "replay": 0.0}
    torch.cuda.synchronize()
    for i in range(60):
        gl = graphs[i % args.lanes]
        t0 = time.perf_counter(); gl.plan.fill_host_(scenes[i % 4])
        t1 = time.perf_counter(); gl.plan.upload_(); gl.static_img.copy_(scenes[i % 4]["img_list"], non_blocking=True)
        t2 = time.perf_counter(); gl.graph.replay()
        t3 = time.perf_counter()
        acc["fill"] += t1 - t0; acc["upload+copy"] += t2 - t1; acc["replay"] += t3 - t2
        if i % 8 == 7:
            torch.cuda.synchronize()
    print("host us per step:", {k: round(v / 60 * 1e6, 1) for k, v in acc.items()}, flush=True)
    for mode in (("one", "streams", "one", "streams") if not args.onegraph else ()):
        run(mode)
        print("%-10s %8.1f depth maps/s" % (mode, run(mode)), flush=True)

    if not args.onegraph:
        return
    # (d) ONE graph holding ``lanes`` scenes as parallel branches, every lane on its own set of auxiliary streams
    from pointmvsnet_amd import pointflow
    lanes = args.onegraph
    plans = [net.make_plan(scenes[l], img_scales, inter_scales, True) for l in range(lanes)]
    imgs = [scenes[l]["img_list"].clone() for l in range(lanes)]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad():
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            sts = [main]
            for l in range(1, lanes):
                pointflow.set_lane(l)
                sts.append(pointflow.side_stream(dev, 9))
                sts[l].wait_stream(main)
            for l in range(lanes):
                pointflow.set_lane(l)
                with torch.cuda.stream(sts[l]):
                    net.run(plans[l], imgs[l], True)
            for st in sts[1:]:
                main.wait_stream(st)
            pointflow.set_lane(0)
    torch.cuda.synchronize()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = args.steps // lanes
        for i in range(n):
            for l in range(lanes):
                plans[l].update_(scenes[(i * lanes + l) % 4])
                imgs[l].copy_(scenes[(i * lanes + l) % 4]["img_list"], non_blocking=True)
            g.replay()
        torch.cuda.synchronize()
        print("onegraph x%d %8.1f depth maps/s" % (lanes, n * lanes / (time.perf_counter() - t0)), flush=True)


if __name__ == "__main__":
    main()
