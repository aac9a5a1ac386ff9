"""Benchmark: depth-maps/sec of the PointFlow path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one whole PointMVSNet.forward(isFlow=True, isTest=True) on one synthetic DTU-like scene of
the named workload (default BASELINE configs[1]: 640x512, 3 views, 48 depth hypotheses, 2 flow
iterations), i.e. one depth map, with both ImageConv towers and VolumeConv inside the timed region and
the inputs resident in HBM when it starts.  Multi-GPU = one process per GPU, scenes sharded round-robin
(pointmvsnet_amd.distributed.shard_scenes), no data-path collective; timing is bracketed by a barrier and
torch.cuda.synchronize() on both sides and the MAX over ranks is used; `value` is the whole-job rate.

``--config cfg4`` times BASELINE configs[3] instead: one training step (forward in train mode, PointMVSNetLoss,
backward, one SUM all-reduce of the flat gradient bucket, RMSprop) per GPU and step, every convolution / BatchNorm /
warp of it on this package's own kernels (pointmvsnet_amd/train_ops.py).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     - the dominant hand-written entry point of the path: achieved = algorithmic bytes (or flops) per
                 launch / average launch duration, measured with HIP events (torch.cuda.Event on the stream the
                 kernels are launched on) around every C-ABI call of an instrumented EAGER calibration pass BEFORE the
                 timed region -- events cannot be recorded inside a replayed hipGraph, so the timed region itself
                 (graph replays, several scene lanes in flight) carries no per-kernel clock; the rocprofv3 kernel
                 trace of the timed execution mode is committed under profiles/ (tools/per_kernel_roofline.py turns it
                 into per-template-instantiation FLOP/s and bytes/s);
  cpu_baseline - the reference's own CPU path when its staged package is present (oracle/_ref, kind "reference"),
                 else the CPU oracle (oracle/pointflow_oracle.py, kind "port"), timed on this host's cores on a
                 bounded sample of the same workload (rank 0, N=1 only);
  kernels      - per-entry-point time split of the calibration pass (informational).
"""
import argparse
import json
import os
import statistics
import sys
import time

PROCESS_T0 = time.perf_counter()          # (the child's experiment budget counts from here: imports included)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointmvsnet_amd import _lib, distributed, synthetic  # noqa: E402
from pointmvsnet_amd.model import PointMVSNet  # noqa: E402

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3         # same guide: dense f32-input MFMA peak (v_mfma_f32_32x32x2 / 16x16x4)


TRAFFIC_TABLE = os.path.join("profiles", "pmc_traffic.json")
UNDER_LOAD_TABLE = os.path.join("profiles", "r06_per_kernel_roofline.json")


PARITY_REPORT = os.path.join("profiles", "r06_parity_report.jsonl")
TRACED_SHA = os.path.join("profiles", "r06_traced_sha.txt")


def traced_sha():
    try:
        return open(os.path.join(ROOT, TRACED_SHA)).read().strip() or None
    except OSError:
        return None


PMC_SHA = os.path.join("profiles", "r06_pmc_sha.txt")


def pmc_sha():
    """The commit the committed PMC passes were taken at (they are not re-taken by every artefact job: the inference kernels
    they count did not change between that commit and the traced one); falls back to the traced commit."""
    try:
        return open(os.path.join(ROOT, PMC_SHA)).read().strip() or traced_sha()
    except OSError:
        return traced_sha()


def parity_summary():
    """Per configuration, from the LAST committed parity report of the GPU test run (tests/conftest.report lines of
    tests/test_gpu_teacher.py and tests/test_gpu_model.py; not measured by this run): the teacher-forced iterations --
    the device against the oracle on the device's own prior and pyramid -- with the oracle's OWN neighbours (worst
    iteration of the configuration: largest relative deviation, fraction of pixels beyond 1e-4, fraction of the pixels
    OUTSIDE every flipped neighbour row's receptive field that deviate by more than 1e-5, which the tests assert to be
    zero), with the device's neighbours fed to the oracle (max norm over every pixel), and the whole forward against the
    reference's golden maps (fraction beyond 1e-4 and largest deviation of the last refined map)."""
    try:
        lines = [json.loads(l) for l in open(os.path.join(ROOT, PARITY_REPORT)) if l.startswith("{")]
    except (OSError, ValueError):
        return None
    out = {}
    for cfg in ("cfg2", "cfg3", "cfg4", "cfg5"):
        own = [r for r in lines if r.get("name", "").startswith("teacher_own_knn_%s" % cfg)]
        same = [r for r in lines if r.get("name", "").startswith("teacher_%s" % cfg) and "rel_max_same_knn" in r]
        if not own and not same:
            continue
        row = {}
        if own:
            row.update(max_rel=max(r["rel_max"] for r in own), frac_gt_1e4=max(r["frac_gt_1e4"] for r in own),
                       frac_outside_fields_gt_1e5=max(r.get("frac_outside_fields_gt_1e5", 0.0) for r in own),
                       max_rel_outside_fields=max(r["rel_max_outside_fields"] for r in own),
                       subgrids_checked=[int(r.get("subgrids_checked", 0)) for r in own],
                       subgrids=[int(r.get("subgrids", 0)) for r in own])
        if same:
            row["max_rel_same_neighbours"] = max(r["rel_max_same_knn"] for r in same)
        out[cfg] = row
    if not out:
        return None
    out["source"] = PARITY_REPORT + " (GPU test run of the traced build %s; committed, not this run)" % (traced_sha() or "?")
    return out


def measured_traffic(entry):
    """HBM bytes per launch of C-ABI entry point `entry` READ FROM THE COMMITTED PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs of an earlier job; tools/pmc_to_traffic.py), or None.  Not measured by this run: the line
    says so in ``roofline.traffic_source``."""
    try:
        table = json.load(open(os.path.join(ROOT, TRAFFIC_TABLE)))
        return float(table["entries"][entry]["bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def under_load_instantiations(prefixes):
    """Best / worst template instantiation of the dominant kernel family UNDER LOAD (four lanes, graph replay), from the
    committed rocprofv3 kernel trace folded by tools/per_kernel_roofline.py -- the calibration clock of this run times
    eager single-chain launches and cannot see them.  None when the table is not there."""
    for name in (UNDER_LOAD_TABLE, os.path.join("profiles", "r06a_per_kernel_roofline.json")):
        try:
            table = json.load(open(os.path.join(ROOT, name)))
        except (OSError, ValueError):
            continue
        # (an instantiation with a fraction of a launch per depth map is a warm-up / calibration launch of another mode)
        rows = [k for k in table.get("kernels", []) if k.get("frac_mfma_peak") and k["kernel"].startswith(prefixes)
                and k.get("launches_per_depth_map", 1.0) >= 0.5]
        if not rows:
            continue
        rows.sort(key=lambda k: k["frac_mfma_peak"])
        us = sum(k["us_per_depth_map"] for k in rows)
        fl = sum(k["flops_per_depth_map"] for k in rows)
        pick = lambda k: {"kernel": k["kernel"], "what": k.get("what"), "frac_mfma_peak": k["frac_mfma_peak"],
                          "us_per_depth_map": k["us_per_depth_map"]}
        return {"source": name + " (rocprofv3 --kernel-trace of the timed execution mode, committed; not this run)",
                "family_us_per_depth_map": us, "family_frac_mfma_peak": fl / us / 1e6 / MFMA_F32_PEAK_TF,
                "best": pick(rows[-1]), "worst": pick(rows[0]), "instantiations": len(rows)}
    return None
WORKLOAD_TEXT = {
    "cfg1": "cfg1: DTU 640x512 (160x128 depth grid), 3 views, 48 hypotheses, 1 flow iter",
    "cfg2": "cfg2: DTU 640x512, 3 source views, 48 depth hypotheses, 2 flow iters",
    "cfg3": "cfg3: DTU 1280x960, 5 views, 96 depth hypotheses, 3 flow iters",
    "cfg5": "cfg5: 1600x1152, 7 views, 96 hypotheses, 3 flow iters (variance aggregation)",
    "cfg4": "cfg4: DTU training step, 640x512, 3 views, 48 hypotheses, 1 scene per GPU, RMSprop + grad all-reduce",
    "cfg5r": "cfg5r: 320x256, 7 views, 16 hypotheses, 3 flow iters (parity fixture only)",
    "tiny": "tiny: 192x128, 3 views, 8 hypotheses, 2 flow iters (smoke only)",
    "small": "small: 320x256, 3 views, 16 hypotheses, 3 flow iters (smoke only)",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node; N > 1 without a torchrun environment re-launches this command under "
                         "torch.distributed.run with N ranks (one process per GPU, RCCL)")
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 40 for the inference configurations: >= 2 s of timed region at cfg 2; "
                         "20 for the training step)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes-per-step", type=int, default=None,
                    help="depth maps per step (default: 64 for cfg1/cfg2, 8 for cfg3/cfg5, 1 for the training step): a "
                         "step is one pass over a batch of that many synthetic scenes, one scene per forward like "
                         "the reference's test.py (TEST.BATCH_SIZE 1), so that the default 20 steps time seconds, "
                         "not 30 ms")
    ap.add_argument("--lanes", type=int, default=4,
                    help="scenes in flight per GPU: that many captured forwards replayed round-robin on as many streams "
                         "(pointmvsnet_amd.graph.LanedForward); 1 = one scene at a time")
    ap.add_argument("--concurrency", type=int, default=None,
                    help="intra-forward concurrency level of eager forwards and of a single lane (default: the "
                         "process default, PF_CONCURRENCY; scene lanes are always single chains, level 0)")
    ap.add_argument("--calibration-steps", type=int, default=10,
                    help="instrumented eager forwards (HIP events around every entry point) before the timed region")
    ap.add_argument("--sync-dir", default=None,
                    help="(route workers) directory of the start barrier: the process writes ready.<pid> after its warm-up "
                         "and starts its timed region when the file `go` appears; the line then carries wall_t0 / wall_t1")
    ap.add_argument("--route-workers", type=int, default=4,
                    help="worker PROCESSES of the drop-in route's second figure (each: the reference's model.py, eager, one "
                         "scene at a time, all on this GPU); 1 = only the one-process figure")
    ap.add_argument("--route", default="fused", choices=["fused", "reference-model"],
                    help="reference-model: time the REFERENCE'S OWN model graph (its unmodified pointmvsnet/model.py, "
                         "--reference-model-py) running eagerly on this package's operator layer "
                         "(compat.load_reference_model) instead of pointmvsnet_amd.model.PointMVSNet")
    ap.add_argument("--reference-model-py", default="/root/reference/pointmvsnet/model.py")
    ap.add_argument("--launch-check", action="store_true",
                    help="only initialise the process group, count the ranks with an all-reduce of ones, print and "
                         "exit (gloo when no GPU is visible: the N > 1 launch path is testable on CPU)")
    ap.add_argument("--config", default="cfg2", choices=sorted(synthetic.CONFIGS))
    ap.add_argument("--eager", action="store_true", help="do not capture the forward in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip value_pcie_inclusive / value_one_lane / route_reference_model / cpu_baseline_cfg1")
    ap.add_argument("--no-train-block", action="store_true",
                    help="cfg2 only: do not time BASELINE configs[3]'s training step after the headline's timed region")
    ap.add_argument("--train-steps", type=int, default=20, help="timed replays of the training step in the train block")
    ap.add_argument("--train-timeout", type=float, default=300.0,
                    help="seconds the train block's child process may take before it is killed (the headline is kept)")
    ap.add_argument("--no-experiments", action="store_true",
                    help="train block: skip its extras (the PF_TRAIN_LAZY_BN=0 arm, the training step's CPU baseline)")
    ap.add_argument("--train-block-only", action="store_true",
                    help="(internal) be the train block's child process: time BASELINE configs[3]'s step, print its JSON")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--train-cpu-baseline", action="store_true",
                    help="cfg4 only: also time ONE oracle training step on the host (tens of seconds, ~15 GB of RAM)")
    return ap.parse_args()


def free_port():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch_command(n, argv):
    """The command ``--gpus N`` re-launches itself as: one process per GPU on this node, rendezvous on 127.0.0.1
    (the form the driver uses; reference: one process driving N GPUs through nn.DataParallel, train.py:177)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def host_threads_per_rank(n_ranks):
    """Host threads a rank may use when ``n_ranks`` processes share this node: its share of the logical CPUs, at most 16
    (the step's host code is a few small NumPy / LAPACK calls; more threads only contend)."""
    return max(1, min(16, int(os.cpu_count() or 1) // max(1, int(n_ranks))))


def maybe_relaunch(args):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: become the launcher.  Under torchrun the
    flag must agree with WORLD_SIZE -- a silent mismatch would print a line for the wrong N."""
    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s; launch with --nproc-per-node %d"
                             % (args.gpus, world, args.gpus))
        return
    if args.gpus <= 1:
        return
    if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d visible" % (args.gpus, torch.cuda.device_count()))
    if not torch.cuda.is_available() and not args.launch_check:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL needs it on this driver
    # N ranks share the host: without a cap every rank starts one OpenMP / ATen thread per logical CPU (N x 128+ threads
    # on the GPU node), which slows the ranks' host code (camera algebra, graph launches) by contention
    env.setdefault("OMP_NUM_THREADS", str(host_threads_per_rank(args.gpus)))
    raise SystemExit(subprocess.call(launch_command(args.gpus, sys.argv[1:]), env=env))


def count_ranks(dev):
    """World size as the COLLECTIVE sees it: an all-reduce(SUM) of ones over RCCL (gloo on CPU)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return 1
    one = torch.ones(1, dtype=torch.float32, device=dev)
    torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
    return int(round(float(one.item())))


def to_device(data, dev):
    out = {k: v.to(dev) for k, v in data.items()}
    out["cam_params_list_host"] = data["cam_params_list"]
    out["mean_host"], out["std_host"] = data["mean"], data["std"]       # host copy: no D2H inside the step
    return out


def _reference_forward():
    """The reference's own PointMVSNet.forward on CPU (reference test.py:58-69,83-84) with the two documented shims of
    tests/golden/make_golden.py: from its tree in the build container, from the byte-for-byte staged copy of its
    hot-path modules on the GPU box (oracle/make_ref.py).  None when neither exists (then the oracle -- an op-for-op
    port, bit-identical to it on every golden -- is what is timed, and ``kind`` says "port")."""
    from oracle import make_ref
    if make_ref.reference_root() is None:     # (/root/reference here; the staged oracle/_ref/pointmvsnet on the GPU box)
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden as MG                                   # applies the shims, imports the reference
        return MG.ref_model
    except Exception:
        return None


def cpu_baseline(net, data, img_scales, inter_scales, repeats, text, train=False, threads=None):
    """CPU baseline on this host's cores: the reference itself where its tree exists, else the oracle (an
    op-for-op port, oracle/pointflow_oracle.py).  torch's default of one thread per logical CPU oversubscribes
    these small operators (128 threads on the GPU box ran 3.8x slower than 8 in the build container), so the
    thread count is swept and the best one is what is reported."""
    from oracle import pointflow_oracle as O
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = None if train else _reference_forward()
    if ref is not None:
        model = ref.PointMVSNet()
        model.load_state_dict(sd)
        model.train()

        def run():
            with torch.no_grad():
                model(data, img_scales, inter_scales, isFlow=True, isTest=True)
    elif train:
        from pointmvsnet_amd.model import PointMVSNetLoss
        loss_fn = PointMVSNetLoss(8.0)
        labels = {"gt_depth_img": synthetic.make_gt_depth(data), "cam_params_list": data["cam_params_list"]}
        names = set(k for k, _ in net.named_parameters())

        def gather_unexpanded(feature, index):
            # the oracle's gather follows the reference (expand to (B, C, N, N), then gather: functions/functions.py:65-67),
            # whose BACKWARD materialises that tensor -- 2.7 TB at 102 400 points; the same values through torch.gather on
            # the unexpanded tensor (autograd: a scatter-add), which is also what makes this baseline runnable at all
            B, C, N = feature.shape
            K = index.shape[2]
            return feature.gather(2, index.reshape(B, 1, N * K).expand(B, C, N * K)).view(B, C, N, K)

        def run():
            leaves = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
            saved = O.gather_knn
            O.gather_knn = gather_unexpanded
            try:
                preds = O.forward(leaves, data, img_scales, inter_scales, True, False)
                sum(loss_fn(preds, labels, True).values()).backward()
            finally:
                O.gather_knn = saved
    else:
        def run():
            with torch.no_grad():
                O.forward(sd, data, img_scales, inter_scales, True, True)
    default_threads = torch.get_num_threads()
    host = int(os.cpu_count() or 1)
    sweep = {}
    try:
        if train:                                                  # one step is tens of seconds: no sweep
            candidates = [min(host, 32)]
        elif threads is not None:                                  # (a second workload: the count already found best)
            candidates = [int(threads)]
            run()
        else:
            candidates = sorted(set(min(host, c) for c in (8, 16, 32, 64, 128)))
            run()                                                  # warm-up (allocator, thread pool)
        for c in candidates:
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            run()
            sweep[c] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        times = [sweep[best]]
        for _ in range(max(0, repeats - 1) if not train else 0):
            t0 = time.perf_counter()
            run()
            times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(default_threads)
    med = statistics.median(times)
    return {"value": 1.0 / med, "unit": "train-scenes/s" if train else "depth-maps/s", "cores": int(best),
            "host_cpus": host, "kind": "reference" if ref is not None else "port",
            "thread_sweep_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": "%d x %s of %s at %d threads, median %.3f s"
                      % (len(times), "training step (forward+loss+backward)" if train else "whole forward", text,
                         best, med)}


# PointFlow-path entry points (the gather path whose algorithmic HBM bytes are tabulated)
GATHER_PATH = ("pf_frustum_variance_f32", "pf_frustum_variance_cl_f32", "pf_nchw_to_nhwc_f32", "pf_flow_pyramid_f32",
               "pf_flow_features_f32", "pf_knn_lattice_f32", "pf_pointwise_gemm_f32", "pf_edge_stats_f32",
               "pf_bn_finalize_jobs_f32", "pf_bn_finalize_f32", "pf_edge_apply_f32", "pf_flow_head_f32")
VOLUME_CONV = ("pf_conv3d_k3_f32", "pf_conv3d_k3_pair_f32", "pf_deconv3d_k3s2_f32", "pf_conv3d_k3_few_f32",
               "pf_conv3d_bottom_f32", "pf_deconv3d_bottom_f32")
VOLUME_CONV_BN = ("pf_channel_bn_apply_f32", "pf_channel_bn_apply2_f32", "pf_channel_stats_f32", "pf_channel_bn_fused_f32")
TOWERS = ("pf_conv2d_wide_sets_f32", "pf_conv2d_wide_f32")
GATHER_PATH_MB = {"cfg1": 404.1, "cfg2": 1658.8, "cfg3": 25194.4, "cfg5": 38116.0}



TRAIN_GROUPS = (
    ("weight_gradients", ("pf_conv_wgrad_f32", "pf_conv_wgrad_batch_f32", "pf_rows_wgrad_f32")),
    ("batchnorm_backward", ("pf_bn_bwd_reduce_f32", "pf_bn_bwd_coeffs_f32", "pf_bn_bwd_apply_f32",
                            "pf_bn_bwd_apply_fused_f32", "pf_rows_bn_bwd_reduce_f32", "pf_rows_bn_bwd_apply_f32")),
    ("warp_backward", ("pf_warp_taps_flow_f32", "pf_warp_taps_frustum_f32", "pf_sort_pairs_by_key",
                       "pf_variance_grad_f32", "pf_warp_gather_f32", "pf_resize_bilinear_backward_f32",
                       "pf_flow_depth_grad_f32")),
    # round 6: two walks over the neighbourhoods (sums, finish) instead of three (reduce, apply + inverse gather)
    ("edgeconv_backward", ("pf_edge_backward_sums_f32", "pf_edge_backward_reduce_f32", "pf_edge_backward_coeffs_f32",
                           "pf_edge_backward_apply_f32", "pf_edge_backward_finish_f32", "pf_knn_inverse")),
)


def train_groups(split, ncal):
    """The weight gradients (f32 MFMA, fixed-order split sums), the BatchNorm backward passes and the warps' backward as
    groups of C-ABI entry points, from the HIP-event calibration of ``ncal`` eager steps."""
    out = {}
    for tag, entries in TRAIN_GROUPS:
        grp = [split[k] for k in entries if k in split]
        if not grp:
            continue
        us = sum(v["ms"] for v in grp) * 1e3 / ncal
        fl = sum(v["flops"] for v in grp) / ncal
        by = sum(v["bytes"] for v in grp) / ncal
        out[tag] = {"kernel_us_per_step": us, "launches_per_step": sum(v["launches"] for v in grp) / float(ncal),
                    "flops_per_step": fl, "algorithmic_bytes_per_step": by,
                    "TFLOPs": fl / us / 1e6 if (us > 0 and fl > 0) else None,
                    "frac_of_f32_mfma_peak": fl / us / 1e6 / MFMA_F32_PEAK_TF if (us > 0 and fl > 0) else None,
                    "GBps": by / us / 1e3 if us > 0 else None}
    return out


def graph_kernel_nodes(make_graphed):
    """Kernel nodes of the captured training step = its dispatches per replay: a second capture with
    ``keep_graph=True`` whose hipGraph is walked with hipGraphGetNodes / hipGraphNodeGetType (the timed capture is the
    ordinary one).  Returns (kernel nodes, all nodes) or (None, reason)."""
    import ctypes
    try:
        graphed = make_graphed(True)
        raw = graphed.graph.raw_cuda_graph()
        # the HIP runtime this process already runs on (torch's bundled one), not whatever "libamdhip64.so" resolves to
        path = None
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
        if path is None:
            return None, "no libamdhip64 mapped"
        hip = ctypes.CDLL(path)
        hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        hip.hipGraphGetNodes.restype = ctypes.c_int
        hip.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        hip.hipGraphNodeGetType.restype = ctypes.c_int
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
            return None, "hipGraphGetNodes failed"
        nodes = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
            return None, "hipGraphGetNodes failed"
        kernels = 0
        for node in nodes:
            kind = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(kind)) == 0 and kind.value == 0:
                kernels += 1                                        # hipGraphNodeTypeKernel
        return kernels, int(n.value)
    except Exception as exc:                                        # measurement aid only: never fail the line for it
        return None, repr(exc)


def train_block(dev, rank, world, steps=20, warmup=3):
    """BASELINE configs[3] beside the headline (after its timed region, headline fields untouched): the captured
    training step -- zero the bucket, forward in train mode, PointMVSNetLoss, backward on this package's own kernels;
    then one SUM all-reduce of the flat bucket and RMSprop -- on one 640x512 scene per GPU (reference train.py:62-67,
    72-86).  ``steps`` replays timed between synchronisations (and barriers when N > 1), MAX over ranks."""
    from pointmvsnet_amd import model as _model
    from pointmvsnet_amd.train_step import GraphedTrainStep, TrainStep
    cfg = os.environ.get("PF_TRAIN_BLOCK_CONFIG", "cfg4")          # ("tiny": the emulator's dry run of this function)
    _, _, _, _, _, img_scales, inter_scales = synthetic.CONFIGS[cfg]
    scenes = []
    for i in range(2):
        data, _, _ = synthetic.make_config(cfg, seed=rank + world * i, train_intrinsics=True)
        batch = to_device(data, dev)
        batch["gt_depth_img"] = synthetic.make_gt_depth(data, seed=rank + world * i).to(dev)
        scenes.append(batch)
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()
    trainer = TrainStep(net)
    # per-entry-point clock: HIP events around every C-ABI call of three EAGER steps (a replay makes no calls)
    for i in range(2):
        trainer(scenes[i % 2], img_scales, inter_scales)
    torch.cuda.synchronize()
    cal = _lib.KernelTimer()
    _lib.set_timer(cal)
    ncal = 3
    from pointmvsnet_amd import train_ops as _train_ops
    wg_streams, _train_ops.WGRAD_STREAMS = _train_ops.WGRAD_STREAMS, 1      # (one stream for the per-entry-point clock: run())
    try:
        for i in range(ncal):
            trainer(scenes[i % 2], img_scales, inter_scales)
    finally:
        _train_ops.WGRAD_STREAMS = wg_streams
    _lib.set_timer(None)
    split = cal.summary()
    own_calls = sum(v["launches"] for v in split.values()) / float(ncal)
    step_flops = sum(v["flops"] for v in split.values()) / ncal
    graphed = GraphedTrainStep(trainer, scenes[0], img_scales, inter_scales)
    for i in range(warmup):
        graphed(scenes[i % 2])
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss, _, _ = graphed(scenes[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    allreduce_us = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(22)]
        ev[0].record()
        for j in range(21):
            trainer.bucket.allreduce_sum()
            ev[j + 1].record()
        torch.cuda.synchronize()
        allreduce_us = statistics.median(ev[j].elapsed_time(ev[j + 1]) for j in range(1, 21)) * 1e3
    assert torch.isfinite(loss)
    ms = elapsed / steps * 1e3
    out = {"metric": "train-scenes/sec (DTU 640x512, 3 views, 1 scene per GPU, forward+loss+backward+all-reduce+RMSprop)"
                     if cfg == "cfg4" else "train-scenes/sec (%s: dry run)" % cfg,
           "value": world * steps / elapsed, "unit": "train-scenes/s", "ms_per_step": ms, "steps": steps,
           "warmup": warmup, "n_gpus": world, "allreduce_us": allreduce_us,
           "execution": "hipGraph replay of zero_grad + forward + loss + backward%s; all-reduce + RMSprop step eager"
                        % (" (flow tower forward / backward on a second stream)" if _model.TRAIN_FORK else ""),
           "dispatches_per_step": None, "dispatches_source": "not counted",
           "own_entry_point_calls_per_step": own_calls,
           "whole_step": {"flops_per_step": step_flops, "TFLOPs": step_flops / (ms / 1e3) / 1e12,
                          "frac_of_f32_mfma_peak": step_flops / (ms / 1e3) / 1e12 / MFMA_F32_PEAK_TF,
                          "entry_point_us_per_step_by_events": sum(v["ms"] for v in split.values()) * 1e3 / ncal},
           "clock": "groups: HIP events around every C-ABI call of %d eager steps (raw pair times; the late weight gradients "
                    "on ONE stream there, on %d in the timed replay); ms_per_step: wall clock over %d graph replays between "
                    "synchronisations" % (ncal, _train_ops.WGRAD_STREAMS, steps)}
    out.update(train_groups(split, ncal))

    def count_dispatches():
        """A second capture with the hipGraph kept, walked for its kernel nodes (after the timing: the caller has already
        published ``out`` when this runs, so whatever happens in here costs only these two fields)."""
        kernels, all_nodes = graph_kernel_nodes(
            lambda keep: GraphedTrainStep(trainer, scenes[0], img_scales, inter_scales, warmup=1, keep_graph=keep))
        if kernels is None:
            out["dispatches_source"] = "unavailable: %s" % (all_nodes,)
            return out
        out["dispatches_per_step"] = kernels + 1 + (1 if world > 1 else 0)
        out["dispatches_source"] = ("kernel nodes of the captured hipGraph (hipGraphGetNodes: %s nodes in all) + RMSprop%s"
                                    % (all_nodes, " + the all-reduce" if world > 1 else ""))
        return out

    return out, count_dispatches

def experiments_block(dev, out, publish):
    """Extras of the train block's CHILD process (each part publishes an updated line when it is done, so a part that dies
    costs only itself):

      lazy_bn       the training step with PF_TRAIN_LAZY_BN=0 (one finalize launch per BatchNorm, round 4's form) beside the
                    default (BatchNorms resolved by their consumers, the backward's rows from one batched finalize; adopted
                    in round 6) -- ms per captured step, same box, and the first step's loss / gradient difference;
      cpu_baseline  the training step's CPU baseline: ONE oracle step on this host's cores.
    (Round 5's bf16x3 tower experiment was measured here on the driver's box -- 1.23-1.42x per layer, +3.0 % on the
    headline: BENCH_r05 -- and removed in round 6: below its own adoption rule of 1.4x.  The PCIe-inclusive arm of the
    headline workload moved into the headline's own process: ``value_pcie_inclusive``.)"""
    from pointmvsnet_amd import train_ops
    from pointmvsnet_amd.train_step import GraphedTrainStep, TrainStep
    exp = out.setdefault("experiments", {})
    # PF_EXPERIMENTS_DRY=1 (the emulator's dry run of this function): the same code on "tiny" scenes, one repetition
    dry = os.environ.get("PF_EXPERIMENTS_DRY") == "1"
    train_cfg, infer_cfg = ("tiny", "tiny") if dry else ("cfg4", "cfg2")
    if dry:
        exp["dry_run"] = True

    def timed_train(lazy):
        train_ops.TRAIN_LAZY_BN = int(lazy)
        _, _, _, _, _, img_scales, inter_scales = synthetic.CONFIGS[train_cfg]
        data, _, _ = synthetic.make_config(train_cfg, seed=0, train_intrinsics=True)
        batch = to_device(data, dev)
        batch["gt_depth_img"] = synthetic.make_gt_depth(data, seed=0).to(dev)
        net = PointMVSNet()
        synthetic.seed_weights(net, seed=0)
        net = net.to(dev).train()
        trainer = TrainStep(net)
        loss, _, _ = trainer(batch, img_scales, inter_scales)
        grad = trainer.bucket.flat.detach().clone()
        graphed = GraphedTrainStep(trainer, batch, img_scales, inter_scales, warmup=1 if dry else 3)
        for _ in range(3):
            graphed(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            graphed(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e3, float(loss), grad

    def timeit(fn, reps=50):
        if dry:
            reps = 1
        for _ in range(1 if dry else 10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / reps

    def part_lazy_bn():
        try:
            ms0, l0, g0 = timed_train(0)
            ms1, l1, g1 = timed_train(1)
            exp["lazy_bn"] = {"ms_per_step_eager_finalize": ms0, "ms_per_step_lazy": ms1, "speedup": ms0 / ms1,
                              "first_step_loss_rel_diff": abs(l1 - l0) / max(abs(l0), 1e-30),
                              "first_step_grad_rel_l2_diff": float((g1 - g0).norm() / g0.norm())}
        except Exception as exc:
            exp["lazy_bn"] = {"error": repr(exc)}
        finally:
            train_ops.TRAIN_LAZY_BN = 1

    def part_cpu_baseline():

        try:          # the training step's CPU baseline: ONE oracle step (forward + loss + backward) on this host's cores
            net = PointMVSNet()
            synthetic.seed_weights(net, seed=0)
            data_cpu, img_scales, inter_scales = synthetic.make_config(train_cfg, seed=0, train_intrinsics=True)
            out["cpu_baseline"] = cpu_baseline(net, data_cpu, img_scales, inter_scales, 1, WORKLOAD_TEXT[train_cfg], train=True)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as exc:
            out["cpu_baseline"] = {"error": repr(exc)}

    # A part that would start after the budget (PF_EXPERIMENTS_BUDGET_S seconds since this process started, default 200)
    # is skipped and says so
    budget = float(os.environ.get("PF_EXPERIMENTS_BUDGET_S", "200"))
    for name, part in (("lazy_bn", part_lazy_bn), ("cpu_baseline", part_cpu_baseline)):
        spent = time.perf_counter() - PROCESS_T0
        if spent > budget:
            exp.setdefault("skipped", []).append("%s (%.0f s of the child's %.0f s budget spent)" % (name, spent, budget))
        else:
            part()
        publish(out)


STAGED_MODEL_PY = os.path.join(ROOT, "oracle", "_ref", "reference_model_py.txt")     # oracle/make_ref.py (git-ignored)


def route_in_child(args):
    """north_star: "pointmvsnet/model.py consumes the new ops unchanged" -- the reference's own model.py (staged byte for
    byte by oracle/make_ref.py) on this package's operator layer, same workload, eager, in a CHILD process (its import
    aliases ``pointmvsnet.*`` to this package, which must not meet the cpu_baseline's import of the reference's own
    modules in one interpreter): depth maps/s, beside -- never in place of -- the headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--route", "reference-model", "--reference-model-py", STAGED_MODEL_PY,
           "--config", args.config, "--no-cpu-baseline", "--no-extras", "--no-train-block", "--steps", "6", "--warmup", "2",
           "--calibration-steps", "2"]
    t0 = time.perf_counter()
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=180)
    except subprocess.TimeoutExpired:
        return {"error": "child killed after 180 s"}
    except OSError as exc:
        return {"error": repr(exc)}
    for line in reversed((proc.stdout or "").splitlines()):
        if line.startswith("{"):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            out = {"value": d["value"], "unit": d["unit"], "ms_per_depth_map": d["ms_per_depth_map"],
                   "execution": "eager: the reference's unmodified model.py on pointmvsnet_amd's operator layer "
                                "(compat.install_as_pointmvsnet), one scene at a time, ONE process",
                   "steps": d["steps"], "scenes_per_step": d["scenes_per_step"],
                   "child_wall_s": time.perf_counter() - t0}
            if args.route_workers > 1:
                out["workers"] = route_workers_in_children(args, cmd, args.route_workers)
            return out
    return {"error": "no result (exit code %s): %s" % (proc.returncode, (proc.stderr or "")[-300:])}


def route_workers_in_children(args, cmd, workers):
    """The drop-in route is HOST-bound (profiles/r06c_route_profile_module_graphs.md: the reference's model.py spends its
    8.5 ms per depth map in its own Python and ATen calls and in the host synchronisations of linspace / inverse / .to();
    the GPU idles most of it) -- the way a deployment fills the GPU with an unmodified model.py is several worker
    PROCESSES per GPU, as the headline route keeps four scene lanes in flight.  `workers` children, each the one-process
    route above (its own interpreter, its own copy of the weights and scenes, all on this GPU), warm up, meet at a start
    barrier (--sync-dir) and run their timed steps together: depth maps of all workers / (last end - first start)."""
    import shutil
    import subprocess
    import tempfile
    sync = tempfile.mkdtemp(prefix="pf_route_")
    t0 = time.perf_counter()
    procs = []
    try:
        for _ in range(workers):
            procs.append(subprocess.Popen(cmd + ["--sync-dir", sync, "--steps", "10"], stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, universal_newlines=True))
        while len([f for f in os.listdir(sync) if f.startswith("ready.")]) < workers:
            if time.perf_counter() - t0 > 120 or any(p.poll() is not None for p in procs):
                raise RuntimeError("a worker did not reach the start barrier")
            time.sleep(0.01)
        open(os.path.join(sync, "go"), "w").close()
        lines = []
        for p in procs:
            so, se = p.communicate(timeout=120)
            got = [l for l in (so or "").splitlines() if l.startswith("{")]
            if not got:
                raise RuntimeError("worker exit code %s: %s" % (p.returncode, (se or "")[-200:]))
            lines.append(json.loads(got[-1]))
        maps = sum(d["steps"] * d["scenes_per_step"] for d in lines)
        span = max(d["wall_t1"] for d in lines) - min(d["wall_t0"] for d in lines)
        return {"processes": workers, "value": maps / span, "unit": "depth-maps/s",
                "per_worker": [round(d["value"], 1) for d in lines],
                "start_skew_ms": round(1e3 * (max(d["wall_t0"] for d in lines) - min(d["wall_t0"] for d in lines)), 2),
                "note": "all workers' depth maps / (last end - first start); every worker is the unmodified model.py, eager",
                "child_wall_s": time.perf_counter() - t0}
    except Exception as exc:
        for p in procs:
            if p.poll() is None:
                p.kill()
        return {"error": repr(exc)}
    finally:
        shutil.rmtree(sync, ignore_errors=True)


def train_block_in_child(args, rank, world):
    """Run train_block in a CHILD process per rank (``--train-block-only``) and return rank 0's JSON (None on the other
    ranks): a crash, an exception on one rank or a hang in there -- the step's RCCL path has never run on more than one
    GPU -- costs the ``train`` entry (it then says what happened), never the headline line.  The children form their own
    process group on a fresh port; a child that outlives ``--train-timeout`` is killed."""
    import subprocess
    env = dict(os.environ)
    if world > 1:
        box = [free_port() if rank == 0 else None]
        torch.distributed.broadcast_object_list(box, src=0)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(box[0])))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # rank 0's child hosts the store of the children's group
    cmd = [sys.executable, os.path.abspath(__file__), "--train-block-only", "--gpus", str(world),
           "--train-steps", str(max(1, args.train_steps)), "--no-cpu-baseline"]
    if getattr(args, "no_experiments", False):
        cmd.append("--no-experiments")
    t0 = time.perf_counter()
    text, note = "", None
    # N > 1 runs no experiments (they are one-GPU A/Bs), so the block is ~40 s of work: a hang in its never-yet-run RCCL
    # path should not cost every N of a scaling sweep the whole one-GPU allowance
    limit = float(args.train_timeout) if world == 1 else min(float(args.train_timeout), 150.0)
    try:
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                              timeout=limit)
        text = proc.stdout or ""
        if proc.returncode != 0:
            note = "child exited with code %s: %s" % (proc.returncode, (proc.stderr or "")[-400:])
    except subprocess.TimeoutExpired as exc:      # what it had printed by then is kept (it publishes part by part)
        text = exc.stdout or ""
        if isinstance(text, bytes):
            text = text.decode("utf8", "replace")
        note = "child killed after %.0f s (--train-timeout)" % limit
    except OSError as exc:
        note = repr(exc)
    if rank != 0:
        return None
    for line in reversed(text.splitlines()):
        if line.startswith("{"):
            try:
                out = json.loads(line)
            except ValueError:
                continue
            out["child_wall_s"] = time.perf_counter() - t0
            if note is not None:
                out["child_note"] = note
            return out
    return {"error": "the train block's child process left no result (%s)" % (note or "no JSON line on its stdout")}


def train_block_child(args, emulate=False):
    """``--train-block-only``: this process IS the child of train_block_in_child."""
    if emulate:                                   # (PF_EMULATE=1: a dry run of this function's code on tests/hipemu)
        out, recount = train_block(torch.device("cpu"), 0, 1, steps=max(1, args.train_steps), warmup=1)
        print(json.dumps(out), flush=True)
        print(json.dumps(recount()), flush=True)
        if not args.no_experiments:
            experiments_block(torch.device("cpu"), out, lambda o: print(json.dumps(o), flush=True))
        return
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS") or host_threads_per_rank(os.environ["WORLD_SIZE"])))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    rank, world, local = distributed.init_from_env()
    dev = torch.device("cuda", local)
    _lib.load()
    recount = None
    try:
        out, recount = train_block(dev, rank, world, steps=max(1, args.train_steps))
    except Exception as exc:                      # (the parent reports it; a rank that fails here may leave the others
        out = {"error": repr(exc)}                # waiting in a collective: the parent's timeout ends that)
    if rank == 0:
        print(json.dumps(out), flush=True)        # the timing first; the parent takes the LAST line it can parse
    if recount is not None and world == 1:        # (one more capture: one rank only, nothing collective in it)
        try:
            out = recount()
            print(json.dumps(out), flush=True)
        except Exception:
            pass
    if world == 1 and "error" not in out and not args.no_experiments:
        try:
            experiments_block(dev, out, lambda o: print(json.dumps(o), flush=True))
        except Exception:
            pass
    if world > 1:
        try:
            torch.distributed.destroy_process_group()
        except Exception:
            pass


def main():
    args = parse_args()
    if os.environ.get("PF_EMULATE") == "1" and not torch.cuda.is_available() and not args.launch_check:
        # development aid for a machine without a GPU (tests/hipemu): the SAME code path -- scenes, model, calibration
        # clock, timed loop, assembly of the JSON line -- with the kernels executed on the host.  Eager, one lane, no
        # child processes; the numbers are the emulator's speed and mean nothing, the line's SHAPE is what gets exercised
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        from on_cpu import emulated_gpu
        args.eager, args.lanes, args.no_cpu_baseline, args.no_train_block, args.gpus = True, 1, True, True, 1
        with emulated_gpu():
            if args.train_block_only:
                return train_block_child(args, emulate=True)
            return run(args, emulate=True)
    return run(args, emulate=False)


def run(args, emulate):
    if args.train_block_only:
        if not torch.cuda.is_available():
            raise RuntimeError("bench.py needs a GPU: the hot path has no CPU fallback")
        return train_block_child(args)
    maybe_relaunch(args)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:               # one of N ranks (torchrun, or the driver's launch line)
        torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS") or host_threads_per_rank(os.environ["WORLD_SIZE"])))
    if args.launch_check:
        rank, world, local = distributed.init_from_env()
        dev = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
        if dev.type == "cuda":
            torch.cuda.set_device(local)
        ranks = count_ranks(dev)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rccl_ranks": ranks,
                              "backend": torch.distributed.get_backend() if world > 1 else None,
                              "host_threads_per_rank": torch.get_num_threads()}))
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    if emulate:
        rank, world, local, dev = 0, 1, 0, torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise RuntimeError("bench.py needs a GPU: the hot path has no CPU fallback")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if local >= torch.cuda.device_count():
            raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPUs visible" % (local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        rank, world, local = distributed.init_from_env()
        dev = torch.device("cuda", local)
    _lib.load()
    rccl_ranks = count_ranks(dev)
    assert rccl_ranks == world == args.gpus, (rccl_ranks, world, args.gpus)

    h, w, V, D, _, img_scales, inter_scales = synthetic.CONFIGS[args.config]
    training = args.config == "cfg4"
    if args.steps is None:
        args.steps = 20 if training else 40
    sps = args.scenes_per_step or {"cfg1": 64, "cfg2": 64, "cfg3": 8, "cfg5": 8, "cfg4": 1}.get(args.config, 8)
    if training:
        sps = 1                                       # a training step is one scene per GPU (BASELINE configs[3])
    total_steps = (args.warmup + args.steps) * sps
    # every rank owns its own scenes (weak scaling: per-GPU work fixed as N grows)
    my_scenes = distributed.shard_scenes(world * total_steps, rank, world)
    # distinct scenes cycled through the timed region: 32 x 11.8 MB of images at cfg 1/2/4 = 377 MB, more than the 256 MB
    # Infinity Cache, so a scene's input is read from HBM (round 5 cycled four scenes, which stayed cache-resident);
    # 16 at cfg 3/5 (74 / 155 MB of images per scene)
    n_unique = min(16 if args.config in ("cfg3", "cfg5") else 32, len(my_scenes))
    scenes = []
    for i in range(n_unique):
        data, _, _ = synthetic.make_config(args.config, seed=my_scenes[i])
        scenes.append(to_device(data, dev))
    if training:                                                                # train intrinsics + ground truth
        scenes = []
        for i in range(n_unique):
            data, _, _ = synthetic.make_config(args.config, seed=my_scenes[i], train_intrinsics=True)
            batch = to_device(data, dev)
            batch["gt_depth_img"] = synthetic.make_gt_depth(data, seed=my_scenes[i]).to(dev)
            scenes.append(batch)
    if args.route == "reference-model":
        from pointmvsnet_amd import compat
        if not os.path.isfile(args.reference_model_py):
            raise SystemExit("bench.py --route reference-model: %s not found" % args.reference_model_py)
        net = compat.load_reference_model(args.reference_model_py).PointMVSNet()
        args.eager, args.lanes = True, 1                     # the reference graph syncs with the host: no capture
        sps = args.scenes_per_step or 4
        scenes = [{k: v for k, v in b.items() if not k.endswith("_host")} for b in scenes]
    else:
        net = PointMVSNet()
    synthetic.seed_weights(net, seed=0)
    net = net.to(dev).train()                                                   # reference test.py:58

    if training:
        from pointmvsnet_amd.train_step import TrainStep
        # PF_MIOPEN_FIND=1: let the library time its convolution solvers (torch.backends.cudnn.benchmark) instead of
        # taking its immediate-mode pick, which for the 3-D weight gradients of VolumeConv is a naive reference kernel
        # (round 4: no library convolution is left in the step -- the knob only matters with train_ops disabled)
        if os.environ.get("PF_MIOPEN_FIND", "0") == "1":
            torch.backends.cudnn.benchmark = True
        trainer = TrainStep(net)
        graphed_train = None
        if not args.eager:
            try:
                from pointmvsnet_amd.train_step import GraphedTrainStep
                graphed_train = GraphedTrainStep(trainer, scenes[0], img_scales, inter_scales)
            except Exception as exc:          # say so, do not hide it
                sys.stderr.write("bench.py: hipGraph capture of the training step failed (%r); running eager\n" % (exc,))
        from pointmvsnet_amd import model as _model
        train_execution = "eager autograd" if graphed_train is None else \
            ("hipGraph replay of zero_grad + forward + loss + backward%s; all-reduce + RMSprop step eager"
             % (" (flow tower forward / backward on a second stream)" if _model.TRAIN_FORK else ""))
        args.eager = True                                                       # (no GraphedForward below)

        def eager_step(i):
            if graphed_train is not None:
                loss, _, preds = graphed_train(scenes[i % n_unique])
            else:
                loss, _, preds = trainer(scenes[i % n_unique], img_scales, inter_scales)
            return preds
    else:
        def eager_step(i):
            with torch.no_grad():
                return net(scenes[i % n_unique], img_scales, inter_scales, isFlow=True, isTest=True)

    step = eager_step

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # ---- eager warm-up, then a calibration pass that times every hand-written entry point --------------
    # HIP events (torch.cuda.Event on the stream the kernels are launched on) around every C-ABI call of
    # ``--calibration-steps`` eager forwards over the same scenes.  This is THE clock of ``roofline`` and ``kernels``:
    # raw event-pair times, nothing subtracted (an empty pair measures ``event_pair_floor_us``, reported beside them;
    # rocprofv3's kernel durations of the same command are committed under profiles/ and are shorter by about that).
    # The pass runs the launch sequence of the timed region: scene lanes are captured as single chains (intra-forward
    # concurrency level 0, where the two towers share their launches), a single lane at the process default.
    from pointmvsnet_amd import pointflow
    if args.concurrency is not None:
        pointflow.CONCURRENCY = int(args.concurrency)
    cal_level = 0 if (not training and not args.eager and args.lanes > 1) else pointflow.CONCURRENCY
    def cal_step(i):
        if training:                    # the C-ABI calls of a step happen in the EAGER step only (a replay makes none)
            return trainer(scenes[i % n_unique], img_scales, inter_scales)[2]
        return eager_step(i)

    with pointflow.concurrency(cal_level):
        for i in range(min(max(args.warmup, 1), 3)):
            cal_step(i)
        torch.cuda.synchronize()
        ncal = max(1, int(args.calibration_steps)) if not training else 3
        cal = _lib.KernelTimer()
        _lib.set_timer(cal)
        # the per-entry-point clock wants launches that do not overlap: the late weight gradients go out on ONE stream for
        # the calibration steps (the timed replay deals them to train_ops.WGRAD_STREAMS streams; two concurrent grids
        # stretch each other's event pairs, which says nothing about either kernel)
        from pointmvsnet_amd import train_ops as _train_ops
        wg_streams, _train_ops.WGRAD_STREAMS = _train_ops.WGRAD_STREAMS, 1
        try:
            for i in range(ncal):
                cal_step(i)
        finally:
            _train_ops.WGRAD_STREAMS = wg_streams
        _lib.set_timer(None)
    split = cal.summary()
    dominant = max(split.items(), key=lambda kv: kv[1]["ms"])[0] if split else None
    by_tag = getattr(cal, "by_tag", {})

    # ---- hipGraph capture of the whole forward (default) ------------------------------------------
    execution = "eager"
    if not args.eager:
        try:
            from pointmvsnet_amd.graph import GraphedForward
            # one captured graph per scene lane; the step's images are copied into the lane's static input (12 MB
            # device-to-device, inside the timed region; one graph per resident scene buffer instead measured slower,
            # profiles/archive/r02/r02ae_graph_slots_ab.log)
            from pointmvsnet_amd.graph import LanedForward
            with torch.no_grad():
                laned = LanedForward(net, scenes[0], img_scales, inter_scales, isFlow=True, isTest=True,
                                     lanes=max(1, args.lanes))
            graphs = laned.graphs

            def step(i):                                                       # noqa: F811
                with torch.no_grad():
                    return laned.submit(scenes[i % n_unique])[1]

            execution = ("hipGraph replay, %d scene lane(s) in flight (per scene: host camera algebra + 1 H2D of the "
                         "scene constants + image copy into the lane's static input + 1 graph launch)" % laned.lanes)
        except Exception as exc:      # capture support varies with the library stack; say so, do not hide it
            sys.stderr.write("bench.py: hipGraph capture failed (%r); running eager\n" % (exc,))
            step = eager_step

    # ---- W untimed warm-up steps in the final execution mode, then the timed region: EXACTLY K steps ------------
    # (event records cannot live inside a replayed graph, so the per-kernel clock is the calibration pass above)
    def run_step(k):
        out = None
        for j in range(sps):
            out = step(k * sps + j)
        return out

    for k in range(args.warmup):
        run_step(k)
    barrier()
    torch.cuda.synchronize()
    if args.sync_dir:                                    # route workers: start together (route_workers_in_children)
        open(os.path.join(args.sync_dir, "ready.%d" % os.getpid()), "w").close()
        t_wait = time.perf_counter()
        while not os.path.exists(os.path.join(args.sync_dir, "go")):
            if time.perf_counter() - t_wait > 150:
                raise SystemExit("bench.py --sync-dir: no `go` within 150 s")
            time.sleep(0.0005)
    wall_t0 = time.time()
    t0 = time.perf_counter()
    for k in range(args.steps):
        preds = run_step(args.warmup + k)
    # wall time until the last step is ENQUEUED (diagnostic only).  Not the host's cost: a lane's pinned constant block is
    # refilled only after its previous upload has run, which queues behind the lane's previous replay, so the enqueue
    # loop is paced by the GPU (one replay queued per lane); the host's own share is ~0.1 ms of camera algebra + ~0.3 ms
    # of copies and graph launch per scene (profiles/archive/r03/r03b_lanes_queues.md)
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    wall_t1 = time.time()
    allreduce_us = None
    if training and world > 1:                   # the step's one collective on its own: 2.8 MB SUM all-reduce over RCCL
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(22)]
        for e0 in ev[:1]:
            e0.record()
        for j in range(21):
            trainer.bucket.allreduce_sum()
            ev[j + 1].record()
        torch.cuda.synchronize()
        allreduce_us = statistics.median(ev[j].elapsed_time(ev[j + 1]) for j in range(1, 21)) * 1e3
    gap_probe = None
    if execution != "eager" and os.environ.get("PF_BENCH_GAP"):      # diagnostic, outside the timed region
        probe = []
        for g in graphs:
            g.probe = probe
        for i in range(min(args.steps * sps, 200)):
            step(i)
        torch.cuda.synchronize()
        for g in graphs:
            g.probe = None
        inside = [a.elapsed_time(b) for a, b in probe]
        between = [probe[i][1].elapsed_time(probe[i + 1][0]) for i in range(len(probe) - 1)]
        gap_probe = {"graph_ms": statistics.median(inside), "between_graphs_ms": statistics.median(between)}
    _lib.set_timer(None)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(preds["flow%d" % len(img_scales)]).all()
    laned_lanes = max(1, args.lanes) if (execution != "eager" and not training) else 1
    lane_probe = [round(r, 1) for r in laned.placement] if (laned_lanes > 1 and laned.placement) else None
    from pointmvsnet_amd import pointflow as _pf
    stage_timeline = _pf.timeline_report() if _pf.TIMELINE else None

    # ---- beside the headline, outside its timed region (never `value`): the same workload (a) with every scene's images
    # starting in PINNED HOST memory -- one asynchronous H2D per scene on the lane's stream before its replay, what the
    # reference's loop pays per scene (test.py:67-71) -- and (b) on ONE scene lane (batch-1 latency mode: one chain in
    # flight, intra-forward forks at the process default)
    value_pcie, value_one_lane, extras_note = None, None, None
    if execution != "eager" and not training and args.route == "fused" and not args.no_extras:
        try:
            n_extra = max(laned.lanes * 8, min(args.steps * sps, 256 if args.config in ("cfg1", "cfg2") else 32))
            hosted = []
            for b in scenes[:min(n_unique, 8)]:
                hb = dict(b)
                hb["img_list"] = b["img_list"].cpu().pin_memory()
                hosted.append(hb)

            def rate(fn, batches, n):
                for i in range(max(4, n // 8)):
                    fn(batches[i % len(batches)])
                barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(n):
                    fn(batches[i % len(batches)])
                torch.cuda.synchronize()
                barrier()
                dt = time.perf_counter() - t1
                if world > 1:
                    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
                    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                    dt = float(tt.item())
                return world * n / dt

            with torch.no_grad():
                value_pcie = rate(laned.submit, hosted, n_extra)
                from pointmvsnet_amd.graph import LanedForward as _LF
                one = _LF(net, scenes[0], img_scales, inter_scales, isFlow=True, isTest=True, lanes=1)
                value_one_lane = rate(one.submit, scenes, max(8, n_extra // 2))
                del one
        except Exception as exc:                       # say so in the line; the headline is already measured
            extras_note = repr(exc)
    with_train = args.config == "cfg2" and args.route == "fused" and not args.no_train_block
    placement_by_rank = None
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, lane_probe)
        placement_by_rank = gathered
    if rank != 0:
        if with_train:                                     # (every rank launches its child of the train block's group)
            train_block_in_child(args, rank, world)
        return
    roof = None
    if dominant is not None:
        s = split[dominant]
        avg_s = s["ms"] / 1e3 / s["launches"]
        avg_bytes = s["bytes"] / s["launches"]
        avg_flops = s["flops"] / s["launches"]
        ridge = MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)              # flop per byte where the roofs meet
        if avg_flops > 0 and avg_flops / max(avg_bytes, 1.0) > ridge:
            achieved = avg_flops / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": dominant, "achieved": achieved, "peak": MFMA_F32_PEAK_TF,
                    "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TF,
                    "algorithmic_flops_per_launch": avg_flops}
        else:
            achieved = avg_bytes / avg_s / 1e9
            roof = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS}
        traffic = measured_traffic(dominant)
        roof.update({"traffic": traffic,
                     "traffic_source": None if traffic is None else
                     TRAFFIC_TABLE + " (two rocprofv3 --pmc passes of a committed earlier job at git %s, FETCH_SIZE / "
                     "WRITE_SIZE; a constant of that job, not measured in this run)" % (pmc_sha() or "?"),
                     "under_load": under_load_instantiations(("conv2d_wide",)) if dominant in TOWERS else None,
                     # the entry point dispatches to one template instantiation per layer shape: the same clock per shape
                     # (eager, single chain -- `under_load` is the same table from the trace of the timed execution mode)
                     "instantiations": sorted(
                         ({"layer": tag, "launches": v["launches"], "avg_launch_us": v["ms"] * 1e3 / v["launches"],
                           "TFLOPs": v["flops"] / (v["ms"] / 1e3) / 1e12 if v["ms"] > 0 else None,
                           "frac_of_f32_mfma_peak": v["flops"] / (v["ms"] / 1e3) / 1e12 / MFMA_F32_PEAK_TF if v["ms"] > 0 else None,
                           "algo_GBps": v["bytes"] / (v["ms"] / 1e3) / 1e9 if v["ms"] > 0 else None}
                          for (entry, tag), v in by_tag.items() if entry == dominant),
                         key=lambda r: -(r["avg_launch_us"] * r["launches"])) or None,
                     "avg_launch_us": avg_s * 1e6,
                     "launches": s["launches"], "algorithmic_bytes_per_launch": avg_bytes,
                     "event_pair_floor_us": s["event_floor_ms"] * 1e3,
                     "clock": "HIP events around every launch of the entry point in %d instrumented eager forwards "
                              "before the timed region (raw pair times, floor not subtracted)" % ncal})
        # north_star: "MFMA utilisation on the 3D-conv path against chip peak" -- VolumeConv's convolution entry
        # points and the two towers' convolutions as groups, same clock (BatchNorm / normalise passes listed apart)
        for tag, entries, bn in (("volume_conv", VOLUME_CONV, VOLUME_CONV_BN), ("towers", TOWERS, ())):
            grp = [split[k] for k in entries if k in split]
            if grp:
                us = sum(v["ms"] for v in grp) * 1e3 / ncal
                fl = sum(v["flops"] for v in grp) / ncal
                roof[tag] = {"flops_per_depth_map": fl, "kernel_us_per_depth_map": us,
                             "launches_per_depth_map": sum(v["launches"] for v in grp) / float(ncal),
                             "TFLOPs": fl / us / 1e6 if us > 0 else None,
                             "frac_of_f32_mfma_peak": fl / us / 1e6 / MFMA_F32_PEAK_TF if us > 0 else None}
                if bn:      # every stand-alone BatchNorm pass of the forward (the towers' materialised stage outputs too)
                    roof[tag]["batchnorm_pass_us_per_depth_map_all"] = sum(
                        split[k]["ms"] for k in bn if k in split) * 1e3 / ncal
    if roof is not None and training:
        # Row Z: the weight gradients (pf_conv_wgrad_f32 / pf_rows_wgrad_f32: f32 MFMA, fixed-order split sums) as a
        # group, the BatchNorm backward passes, and the step's arithmetic as a whole against the f32 matrix peak
        roof.update(train_groups(split, ncal))
        step_flops = sum(v["flops"] for v in split.values()) / ncal
        roof["whole_step"] = {"flops_per_step": step_flops, "ms_per_step": elapsed / args.steps * 1e3,
                              "TFLOPs": step_flops / (elapsed / args.steps) / 1e12,
                              "frac_of_f32_mfma_peak": step_flops / (elapsed / args.steps) / 1e12 / MFMA_F32_PEAK_TF,
                              "entry_point_us_per_step_by_events": sum(v["ms"] for v in split.values()) * 1e3 / ncal}
    kernels = {}
    for k, v in sorted(split.items(), key=lambda kv: -kv[1]["ms"]):
        gbps = (v["bytes"] / (v["ms"] / 1e3) / 1e9) if v["ms"] > 0 else None
        kernels[k] = {"launches_per_depth_map": v["launches"] / float(ncal), "us_per_depth_map": v["ms"] * 1e3 / ncal,
                      "algo_GBps": gbps,
                      "algo_TFLOPs": (v["flops"] / (v["ms"] / 1e3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None}
        if gbps is not None and gbps > HBM_PEAK_GBS:
            # SURVEY 8(d) counts the k-neighbour gather at k*C*4 B per point; those rows are L2 hits (PMC traffic is
            # 6-10x lower), so this figure is an L2-side rate, not a fraction of the HBM roof
            kernels[k]["l2_resident"] = True
            kernels[k]["hbm_traffic_bytes_per_launch"] = measured_traffic(k)
    # the gather path as a whole (north_star: "achieved HBM GB/s on the gather path"): SURVEY 8(d)'s algorithmic
    # bytes per depth map over the summed HIP-event time of the PointFlow entry points in the calibration pass
    gather_ms = sum(v["ms"] for k, v in split.items() if k in GATHER_PATH) / float(ncal)
    if roof is not None and args.config in GATHER_PATH_MB and gather_ms > 0:
        gb = GATHER_PATH_MB[args.config] / 1e3
        roof["gather_path"] = {"algorithmic_MB_per_depth_map": GATHER_PATH_MB[args.config],
                               "kernel_time_ms_per_depth_map": gather_ms,
                               "achieved_GBps": gb / (gather_ms / 1e3),
                               "frac_of_hbm_peak": gb / (gather_ms / 1e3) / HBM_PEAK_GBS,
                               "whole_step_GBps": gb / (elapsed / (args.steps * sps)),
                               "whole_step_frac": gb / (elapsed / (args.steps * sps)) / HBM_PEAK_GBS}
    result = {
        "metric": "depth-maps/sec (DTU 640x512, 3 src views, 2 flow iters)" if args.config == "cfg2"
        else ("train-scenes/sec (DTU 640x512, 3 views, 1 scene per GPU, forward+loss+backward+RMSprop+all-reduce)"
              if training else "depth-maps/sec (%s)" % args.config),
        "value": world * args.steps * sps / elapsed,
        "unit": "train-scenes/s" if training else "depth-maps/s",
        "n_gpus": world,
        "rccl_ranks": rccl_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "scenes_per_step": sps,
        "ms_per_depth_map": elapsed / (args.steps * sps) * 1e3,
        # same workload, same process, outside the timed region; never `value`
        "value_pcie_inclusive": value_pcie,         # images start in pinned host memory: one H2D per scene (test.py:67-71)
        "value_one_lane": value_one_lane,           # one scene in flight (batch-1 latency mode)
        "extras_note": extras_note,
        "unique_scenes_cycled": n_unique,
        "allreduce_us": allreduce_us,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": WORKLOAD_TEXT[args.config], "height": h, "width": w, "views": V, "depth_planes": D,
                   "img_scales": list(img_scales), "inter_scales": list(inter_scales), "batch_per_gpu": 1,
                   "scenes_per_step": sps, "scenes_in_flight_per_gpu": (laned_lanes if not training else 1),
                   "parallelism": ("data parallel x%d, one flat 698 936-float gradient bucket, one SUM all-reduce per step"
                                   % world) if training
                   else "scene-sharded replicas x%d (no data-path collective)" % world,
                   "mode": "train step: forward(isTest=False) + PointMVSNetLoss + backward + RMSprop (train.py:46-112)"
                   if training else
                   "PointMVSNet.forward(isFlow=True, isTest=True), BatchNorm in train mode (test.py:58)"},
        "execution": train_execution if training else execution,
        "route": args.route,
        "wall_t0": wall_t0 if args.sync_dir else None,
        "wall_t1": wall_t1 if args.sync_dir else None,
        "enqueue_wall_ms_per_depth_map": issued / (args.steps * sps) * 1e3,
        "lane_placement_probe_maps_per_s": lane_probe,
        "lane_placement_probe_by_rank": placement_by_rank,
        "host_threads_per_rank": torch.get_num_threads(),
        "gap_probe": gap_probe,
        "stage_timeline_us": stage_timeline,
        "roofline": roof,
        "parity": parity_summary() if not training else None,
        "kernels": kernels,
        "train": None,
    }
    if world == 1 and not args.no_cpu_baseline and (not training or args.train_cpu_baseline):
        data_cpu, _, _ = synthetic.make_config(args.config, seed=my_scenes[0], train_intrinsics=training)
        result["cpu_baseline"] = cpu_baseline(net, data_cpu, img_scales, inter_scales, args.cpu_repeats,
                                              WORKLOAD_TEXT[args.config], train=training)
        result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
        if args.config == "cfg2" and not args.no_extras:
            # BASELINE.json configs[0] ("coarse + 1 flow iter -- reference CPU path (no GPU)"): the same reference forward
            # on cfg 1's workload at the thread count the sweep above found best; a CPU-only figure, nothing on the GPU
            try:
                _, _, _, _, _, sc1, in1 = synthetic.CONFIGS["cfg1"]
                data1, _, _ = synthetic.make_config("cfg1", seed=my_scenes[0])
                result["cpu_baseline_cfg1"] = cpu_baseline(net, data1, sc1, in1, args.cpu_repeats, WORKLOAD_TEXT["cfg1"],
                                                           threads=result["cpu_baseline"]["cores"])
            except Exception as exc:
                result["cpu_baseline_cfg1"] = {"error": repr(exc)}
    else:
        result["cpu_baseline"] = None
    with_route = (args.config == "cfg2" and args.route == "fused" and world == 1 and not args.no_extras
                  and os.path.isfile(STAGED_MODEL_PY))
    if with_train or with_route:
        # the headline line is COMPLETE and on stdout before any child process starts: whatever happens in there (a hang,
        # a GPU fault, the caller's own timeout) cannot cost it.  The final line repeats it with ``train`` /
        # ``route_reference_model`` filled in; a consumer takes the last line it can parse.
        print(json.dumps(result), flush=True)
    if with_train:
        result["train"] = train_block_in_child(args, rank, world)
    if with_route:
        result["route_reference_model"] = route_in_child(args)
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
