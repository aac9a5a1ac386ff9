"""Experiment: scene lanes on CU-MASKED streams (hipExtStreamCreateWithCUMask): does giving every lane its own part of
the chip (own XCDs = own L2s) beat four lanes that share all 256 CUs?

    python tools/exp_cumask.py
"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointmvsnet_amd import pointflow, synthetic  # noqa: E402
from pointmvsnet_amd.graph import GraphedForward, replicate_for_lane  # noqa: E402
from pointmvsnet_amd.model import PointMVSNet  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
NCU = 256


def masked_stream(bits):
    words = (ctypes.c_uint32 * (NCU // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), NCU // 32, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    dev = torch.device("cuda:0")
    _, _, _, _, _, img_scales, inter_scales = synthetic.CONFIGS["cfg2"]
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
    with torch.no_grad(), pointflow.concurrency(0):
        for lane in range(nmax):
            pointflow.set_lane(lane)
            graphs.append(GraphedForward(models[lane], scenes[0], img_scales, inter_scales, warmup=1))
    pointflow.set_lane(0)
    torch.cuda.synchronize()

    def run(streams, steps=240):
        n = len(streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(steps):
                lane = i % n
                with torch.cuda.stream(streams[lane]):
                    graphs[lane](scenes[i % 4])
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0)

    allb = list(range(NCU))
    cases = {
        "4 plain streams": lambda: [torch.cuda.Stream() for _ in range(4)],
        "4 masked, all CUs each": lambda: [masked_stream(allb) for _ in range(4)],
        "4 masked, bits i%8 in {2j,2j+1}": lambda: [masked_stream([b for b in allb if b % 8 in (2 * j, 2 * j + 1)]) for j in range(4)],
        "4 masked, bits [64j,64j+64)": lambda: [masked_stream(list(range(64 * j, 64 * j + 64))) for j in range(4)],
        "4 masked, halves i%8<4 / >=4 (2 lanes each)": lambda: [masked_stream([b for b in allb if (b % 8 < 4) == (j < 2)]) for j in range(4)],
        "4 masked, halves [0,128) / [128,256)": lambda: [masked_stream(list(range(128 * (j // 2), 128 * (j // 2) + 128))) for j in range(4)],
        "8 masked, bit i%8 == j": lambda: [masked_stream([b for b in allb if b % 8 == j]) for j in range(8)],
        "8 masked, bits [32j,32j+32)": lambda: [masked_stream(list(range(32 * j, 32 * j + 32))) for j in range(8)],
        "1 masked, 64 CUs (bits i%8 in {0,1})": lambda: [masked_stream([b for b in allb if b % 8 in (0, 1)])],
        "1 masked, 64 CUs (bits [0,64))": lambda: [masked_stream(list(range(64)))],
        "1 plain stream": lambda: [torch.cuda.Stream()],
    }
    for name, make in cases.items():
        try:
            streams = make()
            run(streams, 48)
            print("%-48s %8.1f" % (name, run(streams)), flush=True)
        except Exception as exc:                       # noqa: BLE001
            print("%-48s failed: %r" % (name, exc), flush=True)


if __name__ == "__main__":
    main()
