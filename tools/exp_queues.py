"""Experiment: which HIP streams share hardware resources?  16 streams are created and first used in order; scene
lanes (single-chain graphs, PF_CONCURRENCY=0) are then placed on chosen subsets of them.

    GPU_MAX_HW_QUEUES=8 PF_CONCURRENCY=0 python tools/exp_queues.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointmvsnet_amd import pointflow, synthetic  # noqa: E402
from pointmvsnet_amd.graph import GraphedForward, replicate_for_lane  # noqa: E402
from pointmvsnet_amd.model import PointMVSNet  # noqa: E402


def main():
    if os.environ.get("EXP_IMPORT_BENCH"):
        import bench  # noqa: F401
    if os.environ.get("EXP_SET_DEVICE"):
        torch.cuda.set_device(0)
    if os.environ.get("EXP_LOAD_LIB"):
        from pointmvsnet_amd import _lib
        _lib.load()
    dev = torch.device("cuda:0")
    _, _, _, _, _, img_scales, inter_scales = synthetic.CONFIGS["cfg2"]
    pool = [torch.cuda.Stream() for _ in range(16)]
    for st in pool:                                        # first use in creation order
        with torch.cuda.stream(st):
            torch.zeros(8, device=dev).add_(1.0)
    torch.cuda.synchronize()
    scenes = []
    for seed in range(4):
        data, _, _ = synthetic.make_config("cfg2", seed=seed)
        b = {k: v.to(dev) for k, v in data.items()}
        b["cam_params_list_host"] = data["cam_params_list"]
        b["mean_host"], b["std_host"] = data["mean"], data["std"]
        scenes.append(b)
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    nmax = 8
    models = [net] + [replicate_for_lane(net) for _ in range(nmax - 1)]
    graphs = []
    with torch.no_grad():
        for lane in range(nmax):
            pointflow.set_lane(lane)
            if os.environ.get("EXP_CAPTURE_ON_LANE"):
                with torch.cuda.stream(pool[lane]):
                    graphs.append(GraphedForward(models[lane], scenes[0], img_scales, inter_scales, warmup=1))
            else:
                graphs.append(GraphedForward(models[lane], scenes[0], img_scales, inter_scales, warmup=1))
    pointflow.set_lane(0)
    torch.cuda.synchronize()

    def run(subset, steps=240):
        n = len(subset)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(steps):
                lane = i % n
                with torch.cuda.stream(pool[subset[lane]]):
                    graphs[lane](scenes[i % 4])
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0)

    subsets = [[0, 1, 2, 3]]
    for sub in subsets:
        run(sub, 60)
        print("%-34s %8.1f" % (sub, run(sub)), flush=True)


if __name__ == "__main__":
    main()
