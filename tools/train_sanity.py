"""End-to-end sanity of the fused training step (BASELINE config 4's shapes): N steps of GraphedTrainStep on a few fixed
synthetic scenes.  Prints the total loss and the allocator's high-water mark every `--every` steps: the loss must fall
(the step really trains: own forward, own backward, flat RMSprop) and the memory must stay flat (replays allocate nothing).

    python tools/train_sanity.py [--steps 200] [--scenes 2] [--lr 1e-4] > gpurun_out/train_sanity.log
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointmvsnet_amd import synthetic  # noqa: E402
from pointmvsnet_amd.model import PointMVSNet  # noqa: E402
from pointmvsnet_amd.train_step import GraphedTrainStep, TrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--scenes", type=int, default=2)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    batches = []
    for i in range(a.scenes):
        data, img_scales, inter_scales = synthetic.make_config("cfg4", train_intrinsics=True, seed=i)
        b = {k: v.to(dev) for k, v in data.items()}
        b["cam_params_list_host"], b["mean_host"], b["std_host"] = data["cam_params_list"], data["mean"], data["std"]
        b["gt_depth_img"] = synthetic.make_gt_depth(data).to(dev)
        batches.append(b)
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    step = GraphedTrainStep(TrainStep(net, lr=a.lr), batches[0], img_scales, inter_scales)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = None
    for i in range(a.steps):
        loss, parts, _ = step(batches[i % a.scenes])
        if i % a.every == 0 or i == a.steps - 1:
            v = float(loss)
            first = v if first is None else first
            print("step %4d  loss %.5f  (%s)  max allocated %.1f MB" % (
                i, v, ", ".join("%s %.4f" % (k, float(x)) for k, x in parts.items()), torch.cuda.max_memory_allocated() / 2 ** 20))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("# %d steps in %.2f s (%.1f steps/s incl. the loss read-backs); loss %.5f -> %.5f" % (a.steps, dt, a.steps / dt, first, v))
    assert v < first, "the loss did not fall"


if __name__ == "__main__":
    main()
