"""Experiment: a two-stage pipeline ACROSS scenes -- stage 1 = both towers + the coarse stage, stage 2 = the PointFlow
iterations -- each stage a captured graph on its own stream, scene i+1's stage 1 beside scene i's stage 2.

    python tools/exp_pipeline.py [--steps 300] [--buffers 2]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointmvsnet_amd import pointflow, synthetic  # noqa: E402
from pointmvsnet_amd.graph import replicate_for_lane  # noqa: E402
from pointmvsnet_amd.model import PointMVSNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--buffers", type=int, default=2)
    ap.add_argument("--split", default="flows", choices=["flows", "tower"])
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
    nb = args.buffers
    nets = [net] + [replicate_for_lane(net) for _ in range(nb - 1)]
    S1, S2 = torch.cuda.Stream(), torch.cuda.Stream()
    plans, imgs, g1s, g2s, outs = [], [], [], [], []
    with torch.no_grad():
        for L in range(nb):
            pointflow.set_lane(L)
            m = nets[L]
            plan = m.make_plan(scenes[L], img_scales, inter_scales, True)
            img = scenes[L]["img_list"].clone()
            for _ in range(2):
                m.run(plan, img, True)
            torch.cuda.synchronize()
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                main_s = torch.cuda.current_stream()
                feats = m.run_coarse_tower(img)
                if args.split == "flows":
                    side = pointflow.side_stream(dev, 0)
                    side.wait_stream(main_s)
                    with torch.cuda.stream(side):
                        pyr = m.run_flow_tower(img)
                    preds = m.run_coarse_stage(plan, feats)
                    main_s.wait_stream(side)
                else:
                    pyr = m.run_flow_tower(img)
                pointflow.flush_counters()
            with torch.cuda.graph(g2, pool=g1.pool()):
                if args.split == "tower":
                    preds = m.run_coarse_stage(plan, feats)
                out = m.run_flows(plan, pyr, preds)
            plans.append(plan); imgs.append(img); g1s.append(g1); g2s.append(g2); outs.append(out)
        pointflow.set_lane(0)
    torch.cuda.synchronize()
    ev1 = [torch.cuda.Event() for _ in range(nb)]
    ev2 = [torch.cuda.Event() for _ in range(nb)]

    def run(n, pipelined):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            L = i % nb
            sc = scenes[i % 4]
            with torch.cuda.stream(S1):
                if i >= nb:
                    S1.wait_event(ev2[L])
                plans[L].update_(sc)
                imgs[L].copy_(sc["img_list"], non_blocking=True)
                g1s[L].replay()
                ev1[L].record(S1)
            st2 = S2 if pipelined else S1
            with torch.cuda.stream(st2):
                st2.wait_event(ev1[L])
                g2s[L].replay()
                ev2[L].record(st2)
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    for mode in (False, True, False, True):
        run(40, mode)
        print("split=%s buffers=%d %-10s %8.1f depth maps/s" % (args.split, nb, "pipelined" if mode else "serial",
                                                                 run(args.steps, mode)), flush=True)
    # sanity: the pipelined result equals a plain forward of the last scene on that buffer
    with torch.no_grad():
        L = (args.steps - 1) % nb
        want = net(scenes[(args.steps - 1) % 4], img_scales, inter_scales, isFlow=True, isTest=True)
    print("last map equals eager:", bool(torch.equal(outs[L]["flow%d" % len(img_scales)], want["flow%d" % len(img_scales)])))


if __name__ == "__main__":
    main()
