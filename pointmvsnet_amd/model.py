"""PointMVSNet model graph on the MI355X operator layer (SURVEY.md section 8, row "glue").

Same class name, constructor, sub-module names (hence state-dict keys), ``forward`` signature and
``preds`` keys as reference ``pointmvsnet/model.py:15-305``; the reference's own ``model.py`` also runs
unchanged on this package's operators (``pointmvsnet_amd.compat.install_as_pointmvsnet``).

Two execution paths:

* inference (no autograd graph needed, the reference's ``test.py`` / validation situation): the fused
  HIP pipeline -- fetch+variance kernel for the cost volume, soft-argmin kernel, and per PointFlow
  iteration one feature-assembly kernel, one lattice-kNN kernel, the GEMM/stats/apply EdgeConv chain
  and the MLP/head kernels, with the r*r test-mode sub-grids batched into single launches;
* training (autograd): the reference's composition on the differentiable HIP operators
  (``FeatureFetcher``, ``gather_knn``) and stock ATen for the rest, so gradients match the reference.

Camera algebra (3x3 inverses, intrinsic scaling) is done once per forward on the host in float32 with
the same ATen CPU calls the reference makes (model.py:54-61, :159-170) and uploaded as small constant
blocks; it is microseconds of work and keeps the geometry bit-identical to the reference's.
"""
import collections
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import networks as _networks
from . import pointflow
from . import train_ops
from .functions.functions import get_pixel_grids, get_propability_map
from .networks import (EdgeConv, EdgeConvNoC, ImageConv, VolumeConv, MAELoss, Valid_MAELoss, tower_pair_supported,
                       tower_pair_views)
from .nn.mlp import SharedMLP
from .utils.feature_fetcher import ChannelLast, FeatureFetcher, frustum_variance
from .utils.torch_utils import get_knn_3d

_HYPOTHESES = (-2, -1, 0, 1, 2)

# 0: the training step differentiates the ATen composition of the conv stacks instead of the train_ops nodes (a module
# attribute the tests flip to compare the two, not a knob)
FUSED_TRAIN = 1

# 1: the training step runs the flow tower (forward AND, through autograd's stream bookkeeping, backward) on a second
# stream beside the coarse stage -- the two are independent until the first PointFlow iteration (reference
# model.py:71-150), and most of a training step's ~700 launches are too small to fill 256 CUs alone.  Inside a captured
# step the fork / join become graph dependencies.  Levels: 0 = one stream (the A/B arm); 1 = the flow tower; 2 = + the
# pyramid-level gradients of the flow-feature nodes (only the flow tower's backward consumes them: off the chain that
# the previous iteration and the coarse stage wait on: +5 % same-box, the default).  (VolumeConv's conv0_1 branch there
# too: -0.4 %, `profiles/r04c_train_streams.md`; not kept.)
TRAIN_FORK = int(os.environ.get("PF_TRAIN_FORK", "2"))
_FORK_STREAMS = {}


def join_fork_streams():
    """The current stream waits for everything the fork streams have been given.  Callers of ``backward()`` on a fused
    training forward run it before they read gradients: autograd joins a side stream at the end of backward only
    where an AccumulateGrad node ran on it, and with ``train_ops.direct_grads()`` the nodes add into the bucket
    themselves."""
    streams = list(_FORK_STREAMS.values())
    train_ops.flush_late(side=streams[0] if streams else None)   # weight gradients that waited for the end of the backward
    for stream in streams:
        torch.cuda.current_stream(stream.device).wait_stream(stream)


def _host_cams(data_batch):
    """Host copy of the camera block.  ``cam_params_list_host`` (optional) spares the D2H copy + sync; it must be
    the same values as the device tensor -- PF_DEBUG=1 checks that on every call (a stale host copy would give
    wrong geometry silently)."""
    cams = data_batch.get("cam_params_list_host")
    if cams is None:
        cams = data_batch["cam_params_list"].detach().cpu()   # one small D2H per forward
    elif os.environ.get("PF_DEBUG") == "1":
        if not torch.equal(cams.float(), data_batch["cam_params_list"].detach().cpu().float()):
            raise RuntimeError("cam_params_list_host differs from cam_params_list")
    return cams.float()


class _Cameras(object):
    """Host-side (float32, ATen-CPU) camera algebra shared by both paths."""

    def __init__(self, cams_host, is_test):
        self.ext = cams_host[:, :, 0, :3, :4].clone()                       # (B,V,3,4)
        self.R = self.ext[:, :, :3, :3]
        self.t = self.ext[:, :, :3, 3].unsqueeze(-1)
        self.R_inv = torch.inverse(self.R)
        self.K_raw = cams_host[:, :, 1, :3, :3].clone()
        K = self.K_raw.clone()
        K[:, :, :2, :3] = K[:, :, :2, :3] / 2.0
        if is_test:
            K[:, :, :2, :3] = K[:, :, :2, :3] / 4.0
        self.K_coarse = K
        self.depth_start = cams_host[:, 0, 1, 3, 0].clone()
        self.depth_interval = cams_host[:, 0, 1, 3, 1].clone()
        self.num_depth = int(cams_host[0, 0, 1, 3, 2].long())
        self.depth_end = self.depth_start + (self.num_depth - 1) * self.depth_interval
        self.is_test = is_test

    def flow_intrinsics(self, image_scale):
        K = self.K_raw.clone()
        K[:, :, :2, :3] *= image_scale if self.is_test else (4 * image_scale)
        return K

    def packed(self, K_flow, mean, std, interval=None):
        """(B, 27 + 21 V + 1) float32: the PF_CAM_* block of include/pointflow_hip.h followed by the
        hypothesis interval of this PointFlow iteration (read by the kernels through a device pointer)."""
        B, V = K_flow.shape[:2]
        kinv = torch.inverse(K_flow[:, 0])
        rows = []
        for b in range(B):
            parts = [kinv[b].reshape(-1), self.R_inv[b, 0].reshape(-1), self.t[b, 0].reshape(-1),
                     mean[b].reshape(-1), std[b].reshape(-1)]
            for v in range(V):
                parts.append(K_flow[b, v].reshape(-1))
                parts.append(self.ext[b, v].reshape(-1))
            parts.append(torch.zeros(1) if interval is None else interval[b].reshape(1))
            rows.append(torch.cat(parts))
        return torch.stack(rows).float().contiguous()


class ScenePlan(object):
    """Every host-derived constant of one forward, in ONE pinned host block mirrored by ONE device block.

    ``update_(data_batch)`` redoes the camera algebra on the host (same ATen-CPU calls as the reference)
    and refreshes the device block with a single asynchronous copy; ``PointMVSNet.run(plan, imgs)`` then
    touches the device only.  That split is what makes the whole forward capturable in a hipGraph and
    replayable on a new scene (pointmvsnet_amd/graph.py), and it replaces ~45 tiny pageable H2D copies per
    depth map by one."""

    def __init__(self, device, B, V, H, W, img_scales, inter_scales, is_test, num_depth):
        self.device, self.B, self.V, self.H, self.W = device, B, V, H, W
        self.img_scales, self.inter_scales = tuple(img_scales), tuple(inter_scales)
        self.is_test, self.D = bool(is_test), int(num_depth)
        for s in self.img_scales:
            if is_test and s not in (0.125, 0.25, 0.5, 1.0):
                raise NotImplementedError
        P = 27 + 21 * V + 1
        self._layout = {}
        off = 0
        for name, shape in [("K_coarse", (B, V, 3, 3)), ("ext", (B, V, 3, 4)), ("Kinv0", (B, 1, 3, 3)),
                            ("Rinv0", (B, 1, 3, 3)), ("t0", (B, 1, 3, 1)), ("depths", (B, self.D)),
                            ("sa_params", (B, 3))] + [("pack%d" % i, (B, P)) for i in range(len(self.img_scales))]:
            n = 1
            for d in shape:
                n *= d
            n_pad = (n + 3) // 4 * 4                       # keep every slice 16-byte aligned
            self._layout[name] = (off, n, shape)
            off += n_pad
        self.host = torch.zeros(off, dtype=torch.float32)
        if torch.cuda.is_available():
            self.host = self.host.pin_memory()
        self._host_np = self.host.numpy()                  # same memory: the NumPy side of fill_host_
        self.dev = torch.zeros(off, dtype=torch.float32, device=device)
        self._copied = None

    def _h(self, name):
        off, n, shape = self._layout[name]
        return self.host[off:off + n].view(shape)

    def d(self, name):
        off, n, shape = self._layout[name]
        return self.dev[off:off + n].view(shape)

    def matches(self, device, B, V, H, W, img_scales, inter_scales, is_test, num_depth):
        return (self.device == device and (self.B, self.V, self.H, self.W) == (B, V, H, W)
                and self.img_scales == tuple(img_scales) and self.inter_scales == tuple(inter_scales)
                and self.is_test == bool(is_test) and self.D == int(num_depth))

    def update_(self, data_batch):
        """Host algebra into the pinned block, then one asynchronous H2D on the current stream."""
        self.fill_host_(data_batch)
        return self.upload_()

    def upload_(self):
        self.dev.copy_(self.host, non_blocking=True)
        if self.dev.is_cuda:
            self._copied = torch.cuda.Event()
            self._copied.record()
        return self

    def fill_host_(self, data_batch):
        """Only the host half: redo the camera algebra into the pinned block (a captured graph that contains the
        H2D copy as its first node reads it at replay time; see graph.GraphedForward).

        The values are ``_Cameras``' (the reference's float32 ATen-CPU arithmetic, model.py:54-61, :159-170), bit for
        bit (tests/test_host.py): the 3x3 inverses are the same LAPACK calls on matrices of the same memory layout
        (two batched ``torch.inverse``), ``torch.linspace`` makes the hypotheses, and what is left -- scaling rows of
        the intrinsics by powers of two, one float32 multiply-add, slicing -- is done on NumPy views straight into the
        pinned block.  ~150 small tensor operations became ~10: 0.72 ms -> ~0.06 ms per scene on the GPU box's host,
        which bounds the scene rate once several scenes are in flight."""
        import numpy as np
        cams_t = _host_cams(data_batch)
        B, V, D = self.B, self.V, self.D
        cams = cams_t.numpy()
        if int(cams[0, 0, 1, 3, 2]) != D:
            raise RuntimeError("ScenePlan: num_depth changed (%d -> %d); build a new plan" % (D, int(cams[0, 0, 1, 3, 2])))
        mean_h = data_batch["mean_host"] if "mean_host" in data_batch else data_batch["mean"].detach().cpu()
        std_h = data_batch["std_host"] if "std_host" in data_batch else data_batch["std"].detach().cpu()
        mean_h = mean_h.float().numpy().reshape(B, 3)
        std_h = std_h.float().numpy().reshape(B, 3)
        f32 = np.float32
        ext = np.ascontiguousarray(cams[:, :, 0, :3, :4])                   # (B,V,3,4), like _Cameras.ext
        K_raw = cams[:, :, 1, :3, :3]
        K_coarse = K_raw.copy()
        K_coarse[:, :, :2, :] /= f32(2.0)
        if self.is_test:
            K_coarse[:, :, :2, :] /= f32(4.0)
        K_flow = []
        for s in self.img_scales:
            K = K_raw.copy()
            K[:, :, :2, :] *= f32(s if self.is_test else 4 * s)
            K_flow.append(K)
        # R^-1 of the reference view: inverted as the strided (row stride 4) view of the (B,V,3,4) block, exactly the
        # operand ``_Cameras`` hands to LAPACK; the intrinsics as packed row-major 3x3
        Rinv0 = torch.inverse(torch.from_numpy(ext)[:, :, :, :3])[:, 0].numpy()                 # (B,3,3)
        if B == 1:                                          # one LAPACK round for every intrinsic matrix of the forward
            kin = np.ascontiguousarray(np.concatenate([K_coarse[:, 0]] + [K[:, 0] for K in K_flow], axis=0))
            kinv = torch.inverse(torch.from_numpy(kin)).numpy()
            Kinv0, Kinv_flow = kinv[:1], [kinv[1 + i:2 + i] for i in range(len(K_flow))]
        else:                                               # (a batch's results depend on the batch's layout: keep _Cameras')
            Kinv0 = torch.inverse(torch.from_numpy(K_coarse)[:, 0]).numpy()
            Kinv_flow = [torch.inverse(torch.from_numpy(K)[:, 0]).numpy() for K in K_flow]
        start, interval = cams[:, 0, 1, 3, 0], cams[:, 0, 1, 3, 1]
        end = start + f32(D - 1) * interval
        t0 = ext[:, 0, :, 3]                                                # (B,3)
        if self._copied is not None:
            self._copied.synchronize()                     # the previous async copy has left the pinned block
        hb, lay = self._host_np, self._layout

        def put(name, arr):
            off, n, _ = lay[name]
            hb[off:off + n] = arr.reshape(-1)

        put("K_coarse", K_coarse)
        put("ext", ext)
        put("Kinv0", Kinv0)
        put("Rinv0", Rinv0)
        put("t0", t0)
        off, n, _ = lay["depths"]
        for b in range(B):
            torch.linspace(float(start[b]), float(end[b]), D, out=self.host[off + b * D:off + (b + 1) * D])
        put("sa_params", np.stack([start, end, interval], axis=1))
        P = 27 + 21 * V + 1
        for i, inter in enumerate(self.inter_scales):
            off, n, _ = lay["pack%d" % i]
            step = f32(inter) * interval                                    # (B,) hypothesis spacing of iteration i
            for b in range(B):
                row = hb[off + b * P:off + (b + 1) * P]
                row[0:9] = Kinv_flow[i][b].reshape(-1)
                row[9:18] = Rinv0[b].reshape(-1)
                row[18:21] = t0[b]
                row[21:24] = mean_h[b]
                row[24:27] = std_h[b]
                per_view = row[27:27 + 21 * V].reshape(V, 21)
                per_view[:, :9] = K_flow[i][b].reshape(V, 9)
                per_view[:, 9:] = ext[b].reshape(V, 12)
                row[P - 1] = step[b]
        return self


class TrainPlan(object):
    """The host-derived constants of one forward of the AUTOGRAD path (reference model.py:45-305), in one pinned
    host block mirrored by one device block -- ScenePlan's idea for the training step: the forward touches device
    tensors only (no per-call H2D copies, no ``.cpu()`` round trip for the flow intrinsics), so forward + loss +
    backward can be captured in a hipGraph and replayed on the next scene (train_step.GraphedTrainStep)."""

    def __init__(self, device, B, V, num_depth, img_scales, inter_scales, is_test):
        self.device, self.B, self.V, self.D = device, B, V, int(num_depth)
        self.img_scales, self.inter_scales = tuple(img_scales), tuple(inter_scales)
        self.is_test = bool(is_test)
        entries = [("K_coarse", (B, V, 3, 3)), ("ext", (B, V, 3, 4)), ("Kinv0", (B, 1, 3, 3)), ("Rinv0", (B, 1, 3, 3)),
                   ("t0", (B, 1, 3, 1)), ("depths", (B, self.D)), ("d_start", (B,)), ("d_int", (B,)),
                   ("mean", (B, 3, 1)), ("std", (B, 3, 1)), ("sa_params", (B, 3))]
        for i in range(len(self.img_scales)):
            entries += [("interval%d" % i, (B,)), ("K_flow%d" % i, (B, V, 3, 3)), ("Kinv_flow%d" % i, (B, 1, 3, 3)),
                        ("pack%d" % i, (B, 27 + 21 * V + 1))]      # the PF_CAM_* block + interval of the fused kernels
        self._layout, off = {}, 0
        for name, shape in entries:
            n = 1
            for d in shape:
                n *= d
            self._layout[name] = (off, n, shape)
            off += (n + 3) // 4 * 4
        self.host = torch.zeros(off, dtype=torch.float32)
        if torch.cuda.is_available():
            self.host = self.host.pin_memory()
        self.dev = torch.zeros(off, dtype=torch.float32, device=device)
        self._copied = None

    def _h(self, name):
        off, n, shape = self._layout[name]
        return self.host[off:off + n].view(shape)

    def d(self, name):
        off, n, shape = self._layout[name]
        return self.dev[off:off + n].view(shape)

    def matches(self, device, B, V, num_depth, img_scales, inter_scales, is_test):
        return (self.device == device and (self.B, self.V, self.D) == (B, V, int(num_depth))
                and self.img_scales == tuple(img_scales) and self.inter_scales == tuple(inter_scales)
                and self.is_test == bool(is_test))

    def fill_host_(self, data_batch):
        cam = _Cameras(_host_cams(data_batch), self.is_test)
        if cam.num_depth != self.D:
            raise RuntimeError("TrainPlan: num_depth changed (%d -> %d); build a new plan" % (self.D, cam.num_depth))
        mean_h = data_batch["mean_host"] if "mean_host" in data_batch else data_batch["mean"].detach().cpu()
        std_h = data_batch["std_host"] if "std_host" in data_batch else data_batch["std"].detach().cpu()
        if self._copied is not None:
            self._copied.synchronize()
        self._h("K_coarse").copy_(cam.K_coarse)
        self._h("ext").copy_(cam.ext)
        self._h("Kinv0").copy_(torch.inverse(cam.K_coarse[:, 0]).unsqueeze(1))
        self._h("Rinv0").copy_(cam.R_inv[:, 0:1])
        self._h("t0").copy_(cam.t[:, 0:1])
        for b in range(self.B):
            self._h("depths")[b].copy_(torch.linspace(float(cam.depth_start[b]), float(cam.depth_end[b]), self.D))
        self._h("d_start").copy_(cam.depth_start)
        self._h("d_int").copy_(cam.depth_interval)
        self._h("sa_params").copy_(torch.stack([cam.depth_start, cam.depth_end, cam.depth_interval], dim=1))
        self._h("mean").copy_(mean_h.float().reshape(self.B, 3, 1))
        self._h("std").copy_(std_h.float().reshape(self.B, 3, 1))
        for i, (s, inter) in enumerate(zip(self.img_scales, self.inter_scales)):
            K_flow = cam.flow_intrinsics(s)
            self._h("interval%d" % i).copy_(inter * cam.depth_interval)
            self._h("K_flow%d" % i).copy_(K_flow)
            self._h("Kinv_flow%d" % i).copy_(torch.inverse(K_flow[:, 0]).unsqueeze(1))
            self._h("pack%d" % i).copy_(cam.packed(K_flow, mean_h.float().reshape(self.B, 3),
                                                   std_h.float().reshape(self.B, 3), inter * cam.depth_interval))
        return self

    def upload_(self):
        self.dev.copy_(self.host, non_blocking=True)
        if self.dev.is_cuda:
            self._copied = torch.cuda.Event()
            self._copied.record()
        return self

    def update_(self, data_batch):
        return self.fill_host_(data_batch).upload_()


class PointMVSNet(nn.Module):
    def __init__(self, img_base_channels=8, vol_base_channels=8, flow_channels=(64, 64, 16, 1), k=16):
        super(PointMVSNet, self).__init__()
        self.k = k
        self.feature_fetcher = FeatureFetcher()
        self.coarse_img_conv = ImageConv(img_base_channels)
        self.coarse_vol_conv = VolumeConv(self.coarse_img_conv.out_channels, vol_base_channels)
        self.flow_img_conv = ImageConv(img_base_channels)
        self.flow_edge_conv = nn.ModuleList([EdgeConvNoC(136, 32), EdgeConv(32, 32), EdgeConv(64, 64)])
        self.flow_mlp = nn.Sequential(
            SharedMLP(32 + 32 * 2 + 64 * 2, flow_channels[:-1]),
            nn.Conv1d(flow_channels[-2], flow_channels[-1], 1, bias=False),
        )
        self._grid_cache = {}
        self._plan = None
        self._tplan = None

    # ------------------------------------------------------------------------------------------
    def _pixel_grid(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._grid_cache:
            self._grid_cache[key] = get_pixel_grids(h, w).to(device)
        return self._grid_cache[key]

    def _hypotheses(self, device):
        key = ("hyp", str(device))                       # cached: a host -> device copy cannot sit inside a hipGraph
        if key not in self._grid_cache:
            self._grid_cache[key] = torch.tensor(_HYPOTHESES, dtype=torch.float32, device=device)
        return self._grid_cache[key]

    def _needs_graph(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    # ------------------------------------------------------------------------------------------
    def make_plan(self, data_batch, img_scales, inter_scales, isTest):
        """Allocate the constant blocks for this shape of problem and fill them from ``data_batch``."""
        img_list = data_batch["img_list"]
        B, V, _, H, W = img_list.shape
        D = int(_host_cams(data_batch)[0, 0, 1, 3, 2].long())
        plan = ScenePlan(img_list.device, B, V, H, W, img_scales, inter_scales, isTest, D)
        return plan.update_(data_batch)

    def forward(self, data_batch, img_scales, inter_scales, isFlow, isTest=False):
        img_list = data_batch["img_list"]
        if not img_list.is_cuda:
            raise RuntimeError("pointmvsnet_amd.PointMVSNet runs on a GPU (HIP) device only; the CPU "
                               "restatement lives in oracle/ and is test infrastructure")
        B, V, _, H, W = img_list.shape
        if self._needs_graph() or B > 1:
            # The fused pipeline is built for one scene per call (the reference's test path asserts
            # TEST.BATCH_SIZE == 1, test.py:116).  With B > 1 every BatchNorm of the PointFlow stage pools its
            # statistics over the batch (networks.py:41,77; nn/conv.py:31-32); the composed path below does
            # exactly that, with or without autograd.
            return self._forward_autograd(data_batch, img_scales, inter_scales, isFlow, isTest)
        D = int(_host_cams(data_batch)[0, 0, 1, 3, 2].long())
        plan = self._plan
        if plan is None or not plan.matches(img_list.device, B, V, H, W, img_scales, inter_scales, isTest, D):
            plan = self._plan = ScenePlan(img_list.device, B, V, H, W, img_scales, inter_scales, isTest, D)
        plan.update_(data_batch)
        return self.run(plan, img_list, isFlow)

    def run(self, plan, img_list, isFlow=True):
        """Device-only inference forward (capturable in a hipGraph): fused HIP pipeline, SURVEY.md section 8.

        The flow tower depends only on the images: it runs on a second stream, beside the coarse stage
        (warp, VolumeConv, soft-argmin -- mostly kernels far too small to fill 256 CUs), and is joined
        before the first PointFlow iteration; under hipGraph capture the fork/join becomes graph edges.
        The fork sits AFTER the coarse tower: two towers side by side only slow each other down (same
        kernels, each fills the chip) -- 474 -> 486 depth maps/s, profiles/archive/r01/r01h_fork_ab.log."""
        dev = img_list.device
        main = torch.cuda.current_stream()
        pointflow.stamp("start")
        if isFlow and pointflow.CONCURRENCY < 1 and tower_pair_supported(self.coarse_img_conv, self.flow_img_conv,
                                                                         img_list):
            # a single chain (one lane of graph.LanedForward): the two towers share their eleven launches
            coarse_cl, pyramids = tower_pair_views(self.coarse_img_conv, self.flow_img_conv, img_list)
            pointflow.stamp("coarse_tower_end")
            preds = self.run_coarse_stage(plan, ChannelLast(coarse_cl))
            return self.run_flows(plan, pyramids, preds)
        feature_list = self.run_coarse_tower(img_list)
        pointflow.stamp("coarse_tower_end")
        if not isFlow:
            preds = self.run_coarse_stage(plan, feature_list)
            pointflow.flush_counters()
            return preds
        if pointflow.CONCURRENCY < 1:
            pyramids = self.run_flow_tower(img_list, raw=True)
            preds = self.run_coarse_stage(plan, feature_list)
            return self.run_flows(plan, pyramids, preds)
        # (fork structure measured in round 2, profiles/archive/r02/r02p_fork_mode_ab.log: flow tower captured first / coarse stage
        # first / fork after the warp / both towers side by side from the first kernel: 598 / 594 / 596 / within 0.3 %)
        side = pointflow.side_stream(dev, 0)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            pyramids = self.run_flow_tower(img_list, raw=True)
            pointflow.stamp("flow_tower_end")
            for p in pyramids.values():
                if isinstance(p, pointflow.RawLevel):
                    for t in (p.raw,) + (tuple(pointflow.affine_rows(p.affine)) if p.affine is not None else ()):
                        t.record_stream(main)
                else:
                    p.record_stream(main)
        preds = self.run_coarse_stage(plan, feature_list)
        main.wait_stream(side)
        return self.run_flows(plan, pyramids, preds)

    # The four device-only stages of ``run`` (GraphedForward may capture them as separate graphs).
    def run_coarse_tower(self, img_list):
        """(B,V,C,FH,FW) coarse features -- or, when the tower's last layer could write them channel-last for the
        warp, a ``ChannelLast`` wrapper around (B,V,FH,FW,C)."""
        out = self.coarse_img_conv.forward_views(img_list, need=("conv3",), channel_last=("conv3",))
        if "conv3_cl" in out:
            return ChannelLast(out["conv3_cl"].contiguous())
        return out["conv3"].contiguous()

    def run_flow_tower(self, img_list, raw=False):
        """The three pyramid levels {"conv1","conv2","conv3"}: (B,V,c,h,w) maps, or with ``raw`` (one scene) the levels
        as ``pointflow.RawLevel`` -- BatchNorm + ReLU pending, applied by the resize kernel of every iteration."""
        if raw and img_list.shape[0] == 1:
            out = self.flow_img_conv.forward_views(img_list, raw=("conv1", "conv2", "conv3"))
            return {n: out[n + "_raw"] if n + "_raw" in out else out[n] for n in ("conv1", "conv2", "conv3")}
        return self.flow_img_conv.forward_views(img_list)

    def run_coarse_stage(self, plan, feature_list):
        """Coarse stage after the tower (reference model.py:79-130): warp, variance, VolumeConv, soft-argmin."""
        B, D = plan.B, plan.D
        preds = collections.OrderedDict()
        if isinstance(feature_list, ChannelLast):
            FH, FW, C = feature_list.maps.shape[2:]
        else:
            C, FH, FW = feature_list.shape[2:]
        # the frustum points (model.py:79-100) are generated inside the fetch+variance kernel
        cost, world_points = frustum_variance(feature_list, plan.d("Kinv0"), plan.d("Rinv0"), plan.d("t0"),
                                              plan.d("depths"), plan.d("K_coarse"), plan.d("ext"))
        pointflow.stamp("warp_end")
        preds["world_points"] = world_points
        filtered = self.coarse_vol_conv.forward_fused(cost.view(B, C, D, FH, FW)).squeeze(1)   # (B,D,FH,FW)
        pred_depth, prob_map = pointflow.soft_argmin_params(filtered, plan.d("sa_params"))
        pointflow.stamp("coarse_stage_end")
        preds["coarse_depth_map"] = pred_depth
        preds["coarse_prob_map"] = prob_map
        return preds

    def run_flows(self, plan, pyramids, preds):
        """Flow stage (reference model.py:132-303) on the coarse depth in ``preds``; flushes the BN counters."""
        B, H, W = plan.B, plan.H, plan.W
        pred_depth = preds["coarse_depth_map"]
        names = ("conv1", "conv2", "conv3")
        for it, img_scale in enumerate(plan.img_scales):
            h, w = int(H * img_scale), int(W * img_scale)
            ratio = int(img_scale * 8) if (plan.is_test and img_scale != 0.125) else 1
            packed = plan.d("pack%d" % it)
            outs, probs = [], []
            for b in range(B):
                pyr_b = [pyramids[n] if isinstance(pyramids[n], pointflow.RawLevel) else pyramids[n][b].contiguous()
                         for n in names]
                d_b, p_b = pointflow.flow_iteration(pyr_b, pred_depth[b, 0], packed[b, -1:], packed[b], h, w, ratio,
                                                    self.flow_edge_conv, self.flow_mlp, k=self.k)
                outs.append(d_b)
                probs.append(p_b)
            pred_depth = torch.stack(outs, dim=0).unsqueeze(1) if B > 1 else outs[0].view(1, 1, h, w)
            flow_prob = torch.stack(probs, dim=0) if B > 1 else probs[0].unsqueeze(0)
            preds["flow{}_prob".format(it + 1)] = flow_prob
            preds["flow{}".format(it + 1)] = pred_depth
            pointflow.stamp("flow%d_end" % (it + 1))
        pointflow.flush_counters()
        return preds

    # ------------------------------------------------------------------------------------------
    def make_train_plan(self, data_batch, img_scales, inter_scales, isTest=False):
        img_list = data_batch["img_list"]
        B, V = img_list.shape[:2]
        D = int(_host_cams(data_batch)[0, 0, 1, 3, 2].long())
        return TrainPlan(img_list.device, B, V, D, img_scales, inter_scales, isTest).update_(data_batch)

    def _forward_autograd(self, data_batch, img_scales, inter_scales, isFlow, isTest, tplan=None):
        """Training path: the reference composition on differentiable HIP operators (model.py:45-305).  Every
        host-derived constant comes from ``tplan`` (a TrainPlan; built and uploaded here when None), so with a
        caller-owned plan the whole pass is device-only and capturable."""
        img_list = data_batch["img_list"]
        dev = img_list.device
        B, V, _, H, W = img_list.shape
        if tplan is None:
            D = int(_host_cams(data_batch)[0, 0, 1, 3, 2].long())
            tplan = self._tplan
            if tplan is None or not tplan.matches(dev, B, V, D, img_scales, inter_scales, isTest):
                tplan = self._tplan = TrainPlan(dev, B, V, D, img_scales, inter_scales, isTest)
            tplan.update_(data_batch)
        return self.run_autograd(tplan, img_list, isFlow)

    def _coarse_cost_autograd(self, feature_list, world_points, K_coarse, ext, D):
        """The reference's coarse cost volume (model.py:102-111) composed from differentiable operators: (B, C, N)."""
        B, V, C = feature_list.shape[:3]
        point_features = self.feature_fetcher(feature_list, world_points, K_coarse, ext)
        ref = feature_list[:, 0].unsqueeze(2).expand(-1, -1, D, -1, -1).contiguous().view(B, C, -1)
        point_features = torch.cat([ref.unsqueeze(1), point_features[:, 1:]], dim=1)
        avg = point_features.mean(dim=1)
        return (point_features ** 2).mean(dim=1) - avg ** 2

    def _fused_train(self, img_list):
        """Row Z: one scene per process and the reference's widths -> every conv / BatchNorm stack, the warps and the
        PointFlow chain are autograd nodes on this package's own forward and backward kernels (train_ops.py)."""
        return bool(FUSED_TRAIN and _networks.FUSED_TRAIN and img_list.shape[0] == 1 and self.training
                    and torch.is_grad_enabled()
                    and train_ops.tower_supported(self.coarse_img_conv, img_list[0])
                    and train_ops.tower_supported(self.flow_img_conv, img_list[0]))

    def run_autograd(self, tplan, img_list, isFlow=True):
        """Device-only autograd forward on the constants of ``tplan`` (capturable together with its backward)."""
        if not self._fused_train(img_list):
            return self._run_autograd(tplan, img_list, isFlow, False)
        # every packed weight of the step (forward layouts, flipped / transposed backward layouts): one launch
        from .train_packs import TrainPacks
        packs = getattr(self, "_train_packs", None)
        if packs is None or packs.stale():
            packs = TrainPacks(self)
            object.__setattr__(self, "_train_packs", packs)
        with torch.cuda.device(img_list.device):
            packs.run()
        side = self._fork_stream(img_list.device) if (TRAIN_FORK and isFlow and img_list.is_cuda) else None
        with train_ops.use_packs(packs), pointflow.deferred_counters(), train_ops.side_stream(side, TRAIN_FORK):
            return self._run_autograd(tplan, img_list, isFlow, True)

    def _fork_stream(self, dev):
        streams = _FORK_STREAMS
        key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=dev)
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)       # intended: a parameter's AccumulateGrad may run beside the stream it was created on
        return streams[key]

    def _run_autograd(self, tplan, img_list, isFlow, fused):
        dev = img_list.device
        B, V, _, H, W = img_list.shape
        isTest, img_scales = tplan.is_test, tplan.img_scales
        preds = collections.OrderedDict()
        K_coarse = tplan.d("K_coarse")
        ext = tplan.d("ext")

        names = ("conv1", "conv2", "conv3")
        levels = side = None
        if fused and isFlow and TRAIN_FORK and img_list.is_cuda:
            main = torch.cuda.current_stream(dev)
            side = self._fork_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                levels = train_ops.tower_train(self.flow_img_conv, img_list[0], names)
        if fused:
            maps3 = train_ops.tower_train(self.coarse_img_conv, img_list[0], ("conv3",))["conv3"]     # (V, C, FH, FW)
            feature_list = maps3.unsqueeze(0)        # (views are used below: an index's backward is a fill + a copy)
        else:
            coarse_maps = [self.coarse_img_conv(img_list[:, v])["conv3"] for v in range(V)]
            feature_list = torch.stack(coarse_maps, dim=1)                   # (B,V,C,FH,FW)
        C, FH, FW = feature_list.shape[2:]
        D = tplan.D
        depths = tplan.d("depths")                                           # (B,D)
        R_inv0 = tplan.d("Rinv0")
        t0 = tplan.d("t0")
        if not (fused and C % 4 == 0 and V <= 8):
            grid = self._pixel_grid(FH, FW, dev).view(1, 1, 3, -1).expand(B, 1, 3, -1)
            uv = torch.matmul(tplan.d("Kinv0"), grid)
            cam_points = (uv.unsqueeze(3) * depths.view(B, 1, 1, D, 1)).view(B, 1, 3, -1)
            world_points = torch.matmul(R_inv0, cam_points - t0).transpose(1, 2).contiguous().view(B, 3, -1)
            preds["world_points"] = world_points

        if fused and C % 4 == 0 and V <= 8:
            # warp + variance as ONE node (forward: the inference kernel; backward: csrc/warp_bwd.hip, no atomics)
            cost, world_points = train_ops.coarse_volume_train(maps3, tplan.d("Kinv0"), R_inv0, t0, depths,
                                                               K_coarse, ext)
            preds["world_points"] = world_points
        else:
            cost = self._coarse_cost_autograd(feature_list, world_points, K_coarse, ext, D)
        cost = cost.view(B, C, D, FH, FW)
        if fused and train_ops.volume_supported(self.coarse_vol_conv, cost):
            filtered = train_ops.volume_train(self.coarse_vol_conv, cost).squeeze(1)
        else:
            filtered = self.coarse_vol_conv(cost).squeeze(1)

        if fused and train_ops.FUSED_HEADS:
            pred_depth, prob_map = train_ops.soft_argmin_train(filtered, tplan.d("sa_params"))
            preds["coarse_depth_map"] = pred_depth
            preds["coarse_prob_map"] = prob_map
        else:
            d_start = tplan.d("d_start")
            d_int = tplan.d("d_int")
            prob_volume = F.softmax(-filtered, dim=1)
            pred_depth = torch.sum(depths.view(B, D, 1, 1) * prob_volume, dim=1).unsqueeze(1)
            preds["coarse_depth_map"] = pred_depth
            preds["coarse_prob_map"] = get_propability_map(prob_volume, pred_depth, d_start, d_int)
        if not isFlow:
            return preds

        if fused:
            if levels is None:
                levels = train_ops.tower_train(self.flow_img_conv, img_list[0], names)
            else:
                main.wait_stream(side)                                       # the join; backward mirrors it
                for n in names:
                    levels[n].record_stream(main)
            # (the fused flow-feature nodes take the level tensors themselves: with a view in between, autograd would add
            # the two iterations' level gradients on the view node's stream, not on the tower's)
            pyramids = {n: levels[n].unsqueeze(0) for n in names}
            pyramids["levels"] = [levels[n] for n in names]
        else:
            per_view = [self.flow_img_conv(img_list[:, v]) for v in range(V)]
            pyramids = {n: torch.stack([pv[n] for pv in per_view], dim=1) for n in names}   # (B,V,c,h_l,w_l)
        if isTest:
            pyramids = {n: ([t.detach() for t in p] if isinstance(p, list) else p.detach()) for n, p in pyramids.items()}
        for it, img_scale in enumerate(img_scales):
            if isTest:
                pred_depth = pred_depth.detach()
                if img_scale not in (0.125, 0.25, 0.5, 1.0):
                    raise NotImplementedError
            h, w = int(H * img_scale), int(W * img_scale)
            ratio = int(img_scale * 8) if (isTest and img_scale != 0.125) else 1
            pred_depth, flow_prob = self._point_flow_autograd(pyramids, pred_depth, tplan, it, h, w, ratio,
                                                              fused=fused and ratio == 1)
            preds["flow{}_prob".format(it + 1)] = flow_prob
            preds["flow{}".format(it + 1)] = pred_depth
        pointflow.flush_counters()
        return preds

    # ------------------------------------------------------------------------------------------
    # differentiable composition (training): reference model.py:150-295 on the HIP operators
    # ------------------------------------------------------------------------------------------
    def _sub_flow_autograd(self, xyz, feature, interval, rows=None):
        """xyz (B,3,D,hs,ws); feature (B,C,D,hs,ws), or ``rows`` (N, C) point-major (one scene, the fused nodes)."""
        B, _, D, hs, ws = xyz.shape
        nn_idx = get_knn_3d(xyz, D, knn=self.k)
        if rows is None:
            x = feature.contiguous().view(B, -1, D * hs * ws)
            if FUSED_TRAIN and _networks.FUSED_TRAIN and B == 1 and self.training:
                rows = x[0].t().contiguous()                                                   # (N, 136) point-major
        else:
            x = None
        if (rows is not None and train_ops.edge_chain_supported(self.flow_edge_conv, rows, nn_idx)
                and len(self.flow_mlp) == 2 and type(self.flow_mlp[1]) is nn.Conv1d
                and self.flow_mlp[1].bias is None and self.flow_mlp[1].out_channels == 1):
            # EdgeConv x3 + the shared MLP as two autograd nodes on point-major rows; the 16 -> 1 convolution and
            # everything after it are a few element-wise ATen operations on (N, 16)
            edges = train_ops.edge_chain_train(self.flow_edge_conv, rows, nn_idx, plane_hw=(hs, ws))   # (N, 224)
            if train_ops.mlp_supported(self.flow_mlp[0], edges):
                act = train_ops.mlp_train(self.flow_mlp[0], edges)                             # (N, 16)
                if train_ops.flow_head_supported(act, self.flow_mlp[1], D) and interval.numel() == 1:
                    # 16 -> 1, softmax over the hypotheses, expected offset: one node (csrc/train_heads.hip)
                    offset, prob = train_ops.flow_head_train(act, self.flow_mlp[1], interval, hs * ws)
                    return offset.view(1, 1, hs, ws), prob.view(1, D, hs, ws)
                flow = (act * self.flow_mlp[1].weight.view(1, -1)).sum(dim=1).view(B, D, hs, ws)
            else:
                flow = self.flow_mlp(edges.t().unsqueeze(0)).contiguous().view(B, D, hs, ws)
        else:
            if x is None:
                x = rows.t().unsqueeze(0)
            edges = []
            for conv in self.flow_edge_conv:
                x = conv(x, nn_idx)
                edges.append(x)
            flow = self.flow_mlp(torch.cat(edges, dim=1)).contiguous().view(B, D, hs, ws)
        prob = F.softmax(-flow, dim=1)
        length = self._hypotheses(xyz.device).view(1, -1, 1, 1) * interval.view(-1, 1, 1, 1)
        return torch.sum(prob * length, dim=1, keepdim=True), prob

    def _point_flow_autograd(self, pyramids, depth_map, tplan, it, h, w, ratio, fused=False):
        dev = depth_map.device
        B = depth_map.shape[0]
        interval, K_flow, ext = tplan.d("interval%d" % it), tplan.d("K_flow%d" % it), tplan.d("ext")
        if depth_map.shape[2] != h:
            depth_map = F.interpolate(depth_map, (h, w), mode="nearest")
        levels = pyramids.get("levels") if (fused and B == 1) else None
        if levels is not None and train_ops.flow_features_supported(levels, depth_map.view(h, w), h, w):
            # feature assembly (5 hypotheses x 3 levels: resize, warp, variance, xyz) as ONE node on the inference kernels,
            # backward without atomics (csrc/warp_bwd.hip); gradients reach the pyramid levels and the prior depth
            pack = tplan.d("pack%d" % it)[0]
            rows, xyz = train_ops.flow_features_train(levels, depth_map.view(h, w), pack[-1:], pack, h, w)
            flow, prob = self._sub_flow_autograd(xyz.view(1, 3, 5, h, w), None, interval, rows=rows)
            return depth_map + flow, prob
        feature, xyz = self._assemble_autograd(pyramids, depth_map, tplan, it, h, w)
        if ratio == 1:
            flow, prob = self._sub_flow_autograd(xyz, feature.view(B, -1, 5, h, w), interval)
        else:
            hs, ws = h // ratio, w // ratio
            f7 = feature.view(B, -1, 5, hs, ratio, ws, ratio)
            x7 = xyz.view(B, 3, 5, hs, ratio, ws, ratio)
            rows_f, rows_p = [], []
            for i in range(ratio):
                cols_f, cols_p = [], []
                for j in range(ratio):
                    fij, pij = self._sub_flow_autograd(x7[:, :, :, :, i, :, j], f7[:, :, :, :, i, :, j], interval)
                    cols_f.append(fij)
                    cols_p.append(pij)
                rows_f.append(torch.stack(cols_f, dim=4))
                rows_p.append(torch.stack(cols_p, dim=4))
            flow = torch.stack(rows_f, dim=3).contiguous().view(B, 1, h, w)
            prob = torch.stack(rows_p, dim=3).contiguous().view(B, 5, h, w)
        return depth_map + flow, prob

    def _assemble_autograd(self, pyramids, depth_map, tplan, it, h, w):
        """The reference's feature assembly (model.py:153-204) composed from differentiable operators: feature
        (B, 136, 5, h*w) and xyz (B, 3, 5, h, w).  depth_map (B, 1, h, w) at the flow resolution."""
        dev = depth_map.device
        B = depth_map.shape[0]
        interval, K_flow, ext = tplan.d("interval%d" % it), tplan.d("K_flow%d" % it), tplan.d("ext")
        mean = tplan.d("mean")
        std = tplan.d("std")
        grid = self._pixel_grid(h, w, dev).view(1, 1, 3, -1).expand(B, 1, 3, -1)
        uv = torch.matmul(tplan.d("Kinv_flow%d" % it), grid)
        R_inv0 = tplan.d("Rinv0")
        t0 = tplan.d("t0")
        resized = {}
        for name, fm in pyramids.items():
            V, c, fh, fw = fm.shape[1:]
            r = F.interpolate(fm.reshape(-1, c, fh, fw), (h, w), mode="bilinear", align_corners=False)
            resized[name] = r.view(B, V, c, h, w)
        feats, xyzs = [], []
        for i in _HYPOTHESES:
            d = depth_map + interval.view(-1, 1, 1, 1) * i
            cam_points = uv * d.view(B, 1, 1, -1)
            world = torch.matmul(R_inv0, cam_points - t0).transpose(1, 2).contiguous().view(B, 3, -1)
            chunks = []
            for name in ("conv1", "conv2", "conv3"):
                pf = self.feature_fetcher(resized[name], world, K_flow, ext)
                avg = pf.mean(dim=1)
                chunks.append((pf ** 2).mean(dim=1) - avg ** 2)
            xyz = (world - mean) / std
            chunks.append(xyz.repeat(1, 8, 1))
            feats.append(torch.cat(chunks, dim=1))
            xyzs.append(xyz)
        feature = torch.stack(feats, dim=2)                                  # (B,136,5,h*w)
        xyz = torch.stack(xyzs, dim=2).contiguous().view(B, 3, 5, h, w)
        return feature, xyz


class PointMVSNetLoss(nn.Module):
    """Coarse + flow1 + flow2 masked MAE, each divided by the number of terms (reference model.py:308-339)."""

    def __init__(self, valid_threshold):
        super(PointMVSNetLoss, self).__init__()
        self.maeloss = MAELoss()
        self.valid_maeloss = Valid_MAELoss(valid_threshold)

    def forward(self, preds, labels, isFlow):
        gt = labels["gt_depth_img"]
        interval = labels["cam_params_list"][:, 0, 1, 3, 1]
        stages = [("coarse_loss", "coarse_depth_map", 1.0)]
        if isFlow:
            stages += [("flow1_loss", "flow1", 0.75), ("flow2_loss", "flow2", 0.375)]
        losses = {}
        n = float(len(stages))
        for name, key, scale in stages:
            pred = preds[key]
            if train_ops.masked_mae_supported(pred, gt, interval):
                # resize + mask + |.| + the sums + the divisions: one launch forward, one backward (csrc/train_heads.hip)
                losses[name] = train_ops.masked_mae(pred, gt, interval, 1.0 / (n * scale))
                continue
            target = F.interpolate(gt, (pred.shape[2], pred.shape[3]))
            losses[name] = self.maeloss(pred, target, scale * interval if scale != 1.0 else interval) / n
        return losses


def _less_pct(pred, gt, interval, threshold, mask):
    err = torch.abs(pred - gt) / interval.view(-1, 1, 1, 1)
    return torch.sum(mask * (err <= threshold).float()) / (torch.sum(mask) + 1e-7)


class PointMVSNetMetric(nn.Module):
    """<1 / <3 interval accuracies per stage (reference model.py:342-420)."""

    def __init__(self, valid_threshold):
        super(PointMVSNetMetric, self).__init__()
        self.valid_threshold = valid_threshold

    def forward(self, preds, labels, isFlow):
        gt = labels["gt_depth_img"]
        interval = labels["cam_params_list"][:, 0, 1, 3, 1]
        coarse = preds["coarse_depth_map"]
        target = F.interpolate(gt, (coarse.shape[2], coarse.shape[3]))
        valid = (target != 0.0).float()
        metrics = {"<1_pct_cor": _less_pct(coarse, target, interval, 1.0, valid),
                   "<3_pct_cor": _less_pct(coarse, target, interval, 3.0, valid)}
        if isFlow:
            before = coarse
            for name, scale in (("flow1", 0.75), ("flow2", 0.375)):
                pred = preds[name]
                target = F.interpolate(gt, (pred.shape[2], pred.shape[3]))
                iv = scale * interval
                if before.size(2) != pred.size(2):
                    before = F.interpolate(before, (pred.shape[2], pred.shape[3]))
                prior_err = torch.abs(before - target) / iv.view(-1, 1, 1, 1)
                mask = (prior_err < self.valid_threshold).float() * (target != 0.0).float()
                metrics["<1_pct_" + name] = _less_pct(pred, target, iv, 1.0, mask)
                metrics["<3_pct_" + name] = _less_pct(pred, target, iv, 3.0, mask)
                before = pred
        return metrics


def build_pointmvsnet(cfg):
    net = PointMVSNet(img_base_channels=cfg.MODEL.IMG_BASE_CHANNELS,
                      vol_base_channels=cfg.MODEL.VOL_BASE_CHANNELS,
                      flow_channels=cfg.MODEL.FLOW_CHANNELS)
    return net, PointMVSNetLoss(valid_threshold=cfg.MODEL.VALID_THRESHOLD), \
        PointMVSNetMetric(valid_threshold=cfg.MODEL.VALID_THRESHOLD)
