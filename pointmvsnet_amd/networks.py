"""Operator layer consumed by the model graph.

``EdgeConv`` / ``EdgeConvNoC`` / ``ImageConv`` / ``VolumeConv`` keep the reference's class names,
constructor signatures, forward signatures and parameter / buffer names (reference networks.py:9-167),
so reference checkpoints load and reference ``model.py`` runs on them unchanged.

EdgeConv semantics are the reference's CUDA branch (neighbours are gathered from ``conv2``'s output,
networks.py:26-28).  Two execution paths, both on HIP kernels:

* inference (no autograd graph needed): the fused GEMM / stats / apply kernels of
  ``pointmvsnet_amd.pointflow`` -- the (B,C,N,k) edge tensor is never materialised;
* training (autograd enabled and something requires grad): the reference's own composition --
  1x1 convs, the HIP ``gather_knn`` (forward and scatter-add backward kernels), BatchNorm2d, ReLU, mean --
  so gradients flow exactly as in the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import pointflow
from .functions.gather_knn import gather_knn
from .nn.conv import *  # noqa: F401,F403  (the reference re-exports the conv blocks from here)
from .nn.conv import Conv2d, Conv3d, Deconv3d


class _EdgeConvTrain(torch.autograd.Function):
    """EdgeConv / EdgeConvNoC with a train-mode BatchNorm as ONE autograd node on the fused HIP kernels.

    Forward = the inference kernels (GEMM, pair statistics, apply); saved for backward: the input, the rows
    [l | e] and per-channel statistics -- not the (B,2C,N,k) edge tensor the reference's autograd keeps
    (networks.py:18-45: 839 MB for the 102 400-point lattice of BASELINE config 4).  Backward recomputes
    d = e[idx] - l in two gather passes (BatchNorm reduction terms, then dl / scatter-added de), followed by
    two plain GEMMs for dX and dW."""

    @staticmethod
    def forward(ctx, feature, knn_inds, w1, w2, gamma, beta, bn, concat):
        B, cin, N = feature.shape
        C = w1.shape[0]
        k = knn_inds.shape[2]
        width = (2 if concat else 1) * C
        x = feature.detach().float().contiguous()
        idx = knn_inds.contiguous()
        out = torch.empty((B * N, width), dtype=torch.float32, device=x.device)
        keep = {}
        with _lib.on_device(x.device):
            pointflow.edge_conv_fused(x, False, 0, cin, B, N, idx, w1, w2, bn, concat, out, width,
                                      groups_per_stat=B, keep=keep)
            pointflow.flush_counters()
        # everything the backward recomputes from goes through save_for_backward (in-place modification checks, and a
        # second backward -- retain_graph, checkpoint re-entrance -- finds the tensors again)
        ctx.save_for_backward(x, idx, w1, w2, keep["LE"], keep["scale"], keep["shift"], keep["mean"], keep["invstd"])
        ctx.meta = (bool(concat), C, k, B, N, cin)
        return out.view(B, N, width).transpose(1, 2).contiguous()

    @staticmethod
    def backward(ctx, grad_out):
        x, idx, w1, w2, LE, scale, shift, mean, invstd = ctx.saved_tensors
        keep = {"LE": LE, "scale": scale, "shift": shift, "mean": mean, "invstd": invstd}
        concat, C, k, B, N, cin = ctx.meta
        width = (2 if concat else 1) * C
        gy = grad_out.float().transpose(1, 2).contiguous().view(B * N, width)        # point-major rows
        with _lib.on_device(x.device):
            grad_le, grad_gamma, grad_beta = pointflow.edge_conv_backward(keep, idx, gy, C, k, B, N, B, concat)
        dle = grad_le.view(B, N, 2 * C)
        wcat = torch.cat([w1.detach().reshape(C, cin), w2.detach().reshape(C, cin)], dim=0).float()   # (2C, K)
        grad_x = torch.matmul(wcat.t(), dle.transpose(1, 2))                          # (B, K, N)
        grad_w = torch.matmul(dle.transpose(1, 2), x.transpose(1, 2)).sum(dim=0)      # (2C, K)
        return (grad_x, None, grad_w[:C].reshape(w1.shape), grad_w[C:].reshape(w2.shape), grad_gamma, grad_beta,
                None, None)


# 0: training runs the reference's composition (gather_knn + BatchNorm2d + ...) instead of the fused autograd node
# (a module attribute, not an environment switch: the tests compare the two)
FUSED_TRAIN = 1


class _EdgeConvBase(nn.Module):
    concat = True

    def __init__(self, in_channels, out_channels):
        super(_EdgeConvBase, self).__init__()
        self.conv1 = nn.Conv1d(in_channels, out_channels, 1, bias=False)
        self.conv2 = nn.Conv1d(in_channels, out_channels, 1, bias=False)
        self.bn = nn.BatchNorm2d((2 if self.concat else 1) * out_channels)

    def _needs_graph(self, feature):
        if not torch.is_grad_enabled():
            return False
        return feature.requires_grad or any(p.requires_grad for p in self.parameters())

    def _forward_autograd(self, feature, knn_inds):
        k = knn_inds.shape[2]
        local_feature = self.conv1(feature)
        edge_feature = self.conv2(feature)
        neighbour = gather_knn(edge_feature, knn_inds)                       # HIP fwd + HIP bwd
        central = local_feature.unsqueeze(-1).expand(-1, -1, -1, k)
        diff = neighbour - central
        edge = torch.cat([central, diff], dim=1) if self.concat else diff
        edge = F.relu(self.bn(edge), inplace=True)
        return torch.mean(edge, dim=3)

    def _forward_fused(self, feature, knn_inds):
        B, cin, N = feature.shape
        cout = self.conv1.weight.shape[0]
        width = (2 if self.concat else 1) * cout
        x = feature.detach().float().contiguous()
        idx = knn_inds.contiguous()
        out = torch.empty((B * N, width), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            pointflow.edge_conv_fused(x, False, 0, cin, B, N, idx, self.conv1.weight, self.conv2.weight,
                                      self.bn, self.concat, out, width, groups_per_stat=B)
            pointflow.flush_counters()
        return out.view(B, N, width).transpose(1, 2).contiguous()

    def forward(self, feature, knn_inds):
        _lib.require_gpu(feature, knn_inds)
        if feature.dim() != 3 or knn_inds.dim() != 3 or knn_inds.dtype != torch.int64:
            raise RuntimeError("EdgeConv: expected feature (B,C,N) and int64 knn_inds (B,N,k)")
        if self._needs_graph(feature):
            if (FUSED_TRAIN and self.bn.training and self.bn.momentum is not None and self.bn.affine
                    and self.conv1.weight.shape[0] in (32, 64) and knn_inds.shape[2] >= 1):
                return _EdgeConvTrain.apply(feature, knn_inds, self.conv1.weight, self.conv2.weight,
                                            self.bn.weight, self.bn.bias, self.bn, self.concat)
            return self._forward_autograd(feature, knn_inds)
        if self.conv1.weight.shape[0] not in (32, 64) or knn_inds.shape[2] < 1:
            # widths the fused kernels are not built for (their GEMM emits [l | e] = 2*C_out <= 128 columns):
            # the composed operators give the same result, like the reference, instead of an error
            with torch.no_grad():
                return self._forward_autograd(feature, knn_inds)
        if feature.dtype == torch.float32:
            from . import graph
            return graph.module_forward(self, self._forward_fused, feature, knn_inds)
        return self._forward_fused(feature, knn_inds)


class EdgeConv(_EdgeConvBase):
    """(B,C_in,N) -> (B, 2*C_out, N): mean_k ReLU(BN(cat[l, e[idx] - l]))."""
    concat = True


class EdgeConvNoC(_EdgeConvBase):
    """(B,C_in,N) -> (B, C_out, N): mean_k ReLU(BN(e[idx] - l))."""
    concat = False


def _deconv_block_ok(block):
    conv = getattr(block, "conv", None)
    return (type(conv) is nn.ConvTranspose3d and conv.kernel_size == (3, 3, 3) and conv.stride == (2, 2, 2)
            and conv.padding == (1, 1, 1) and conv.output_padding == (1, 1, 1) and conv.dilation == (1, 1, 1)
            and conv.groups == 1 and conv.bias is None)


def _deconv_fusable(block, x):
    """pf_deconv3d_k3s2_f32 covers the decoder's ConvTranspose3d blocks; measured policy: from 2048 input
    cells up (below that the layer is a handful of wavefronts either way and the library GEMM wins)."""
    return _deconv_block_ok(block) and x[0, 0].numel() >= 2048


def _conv3d_fusable(conv, x):
    """Layers pf_conv3d_k3_f32 takes (measured policy, profiles/archive/r01/r01l_microbench_conv3d.log: 16->32 /2 on
    24x32x40: 14.5 us vs 23.4 us for the library's im2col + GEMM, 32->32 on 12x16x20: 19.6 vs 36.5 us; the
    6x8x10 layers stay on the library)."""
    return (type(conv) is nn.Conv3d and conv.kernel_size == (3, 3, 3) and conv.padding == (1, 1, 1)
            and conv.stride in ((1, 1, 1), (2, 2, 2)) and conv.dilation == (1, 1, 1) and conv.groups == 1
            and conv.bias is None and conv.in_channels % 4 == 0 and conv.out_channels <= 32
            and x[0, 0].numel() // (conv.stride[0] ** 3) >= 2048)


def _block_fused(block, x, samples_per_stat):
    """conv (library) -> HIP BatchNorm statistics/finalize -> HIP affine+ReLU in place; ``block`` is one of
    the nn.conv blocks or a plain nn.ConvNd (no BN / ReLU)."""
    skip = None
    if isinstance(x, tuple):                  # decoder input "up + skip" (reference networks.py:163-165)
        x, skip = x
    if (skip is not None or hasattr(block, "bn")) and _deconv_fusable(block, x):
        # transposed conv with the skip add on load and the BN batch statistics in the epilogue
        training_bn = block.bn is not None and (block.bn.training or not block.bn.track_running_stats)
        y, partials = pointflow.deconv3d_k3s2(x.contiguous(), None if skip is None else skip.contiguous(),
                                              block.conv.weight, training_bn)
        if block.bn is not None:
            return pointflow.batch_norm_act_(y, block.bn, block.relu, samples_per_stat, partials=partials)
        return F.relu(y, inplace=True) if block.relu else y
    if not hasattr(block, "bn"):
        if (type(block) is nn.Conv3d and block.kernel_size == (3, 3, 3) and block.padding == (1, 1, 1)
                and block.stride == (1, 1, 1) and block.dilation == (1, 1, 1) and block.groups == 1
                and block.bias is None and block.out_channels <= 4 and block.in_channels * block.out_channels <= 256):
            return pointflow.conv3d_k3_few((x if skip is None else x + skip).contiguous(), block.weight)
        return block(x if skip is None else x + skip)
    if skip is not None:
        x = x + skip
    conv = block.conv
    training_bn = block.bn is not None and (block.bn.training or not block.bn.track_running_stats)
    if _conv3d_fusable(conv, x):
        # row R on the f32 matrix cores; the BN batch statistics come out of the conv epilogue
        y, partials = pointflow.conv3d_k3(x.contiguous(), conv.weight, conv.stride[0], training_bn)
        if block.bn is not None:
            return pointflow.batch_norm_act_(y, block.bn, block.relu, samples_per_stat, partials=partials)
        return F.relu(y, inplace=True) if block.relu else y
    pointflow.warn_library_fallback("VolumeConv layer", conv)
    y = block._crop(conv(x), x)
    if block.bn is not None:
        return pointflow.batch_norm_act_(y.contiguous(), block.bn, block.relu, samples_per_stat)
    return F.relu(y, inplace=True) if block.relu else y


class ImageConv(nn.Module):
    """2D feature tower: strides 1/2/2/2, widths b/2b/4b/8b; returns {"conv0".."conv3"}."""

    def __init__(self, base_channels):
        super(ImageConv, self).__init__()
        b = base_channels
        self.base_channels = b
        self.out_channels = 8 * b

        def stage(cin, cout, last_plain=False):
            tail = nn.Conv2d(cout, cout, 3, padding=1, bias=False) if last_plain \
                else Conv2d(cout, cout, 3, 1, padding=1)
            return nn.Sequential(Conv2d(cin, cout, 5, stride=2, padding=2),
                                 Conv2d(cout, cout, 3, 1, padding=1), tail)

        self.conv0 = nn.Sequential(Conv2d(3, b, 3, 1, padding=1), Conv2d(b, b, 3, 1, padding=1))
        self.conv1 = stage(b, 2 * b)
        self.conv2 = stage(2 * b, 4 * b)
        self.conv3 = stage(4 * b, 8 * b, last_plain=True)

    def forward(self, imgs):
        """The reference's call (model.py:71-77, :140-148: one view at a time, BatchNorm statistics over the batch).
        Without an autograd graph the HIP tower kernels run (the batched-views path with one view); with one, the
        stock ATen composition, whose backward the training step uses."""
        if pointflow.hip_inference(imgs, self):
            def run(x):
                out = self.forward_views(x.unsqueeze(1), need=("conv0", "conv1", "conv2", "conv3"))
                pointflow.flush_counters()
                return {k: v[:, 0] for k, v in out.items()}
            from . import graph
            return graph.module_forward(self, run, imgs)
        out = {}
        x = imgs
        for name in ("conv0", "conv1", "conv2", "conv3"):
            x = getattr(self, name)(x)
            out[name] = x
        return out

    def forward_views(self, img_list, need=("conv1", "conv2", "conv3"), channel_last=(), raw=()):
        """Inference fast path: all V views of (B,V,3,H,W) in ONE pass through the tower with per-view
        BatchNorm statistics -- numerically the reference's V separate calls (model.py:71-77).  Each layer is
        one pf_conv2d_wide_f32 launch (previous BatchNorm+ReLU applied while staging, this layer's statistics in
        the epilogue) plus the finalize; only the stage outputs in ``need`` ((B,V,c,h,w); the coarse tower
        needs "conv3" alone) are materialised, every other BatchNorm+ReLU stays an affine row pair that the
        next convolution applies while staging -- "conv0" is never returned.  Stage names in ``channel_last`` come
        back as (B,V,h,w,c) under the key name + "_cl" when the stage's last layer is a plain convolution on the
        wide kernel (the coarse tower's "conv3" feeds the channel-last warp: no transposition pass), else as usual.
        Stage names in ``raw`` (B = 1) come back under name + "_raw" as ``pointflow.RawLevel``: the stage's last
        convolution output (V,c,h,w) with its BatchNorm + ReLU still pending as (scale, shift) rows -- the flow
        tower's three levels are normalised by the kernel that resizes them (pf_flow_pyramid_f32), not by a pass of
        their own."""
        B, V = img_list.shape[:2]
        x = img_list.transpose(0, 1).reshape(V * B, *img_list.shape[2:]).float().contiguous()   # view-major
        pending = None                      # (scale, shift) of a BatchNorm+ReLU not yet applied to x
        blocks = [(name, blk) for name in ("conv0", "conv1", "conv2", "conv3") for blk in getattr(self, name)]
        out = {}
        for i, (name, block) in enumerate(blocks):
            stage_end = i + 1 == len(blocks) or blocks[i + 1][0] != name
            nxt = blocks[i + 1][1] if i + 1 < len(blocks) else None
            # the BN+ReLU of this block can stay pending only if the next conv applies it while staging
            nconv = None if nxt is None else (nxt.conv if hasattr(nxt, "bn") else nxt)
            wanted = stage_end and name in need
            as_raw = wanted and name in raw and B == 1 and (nconv is None or pointflow.conv2d_wide_preferred(nconv))
            defer = (as_raw or not wanted) and nconv is not None and pointflow.conv2d_wide_preferred(nconv)
            lazy = bool(defer) and not as_raw      # the next conv resolves this BatchNorm (a raw level needs the rows)
            conv = block.conv if hasattr(block, "bn") else block
            cl = (wanted and name in channel_last and not hasattr(block, "bn") and conv.out_channels >= 32
                  and pointflow.conv2d_wide_preferred(conv))
            x, pending = _conv2d_block_fused(block, x, pending, B, defer, lazy, channel_last_out=cl)
            if as_raw:
                out[name + "_raw"] = pointflow.RawLevel(x, pending)
            elif wanted:
                out[name + "_cl" if cl else name] = x.view(V, B, *x.shape[1:]).transpose(0, 1)
        return out


def tower_pair_supported(coarse, flow, img_list):
    """Both towers on the shared launches of ``tower_pair_views``: one scene, identical towers whose every layer is
    a tower-kernel shape, train-mode BatchNorm everywhere (the reference's test mode, test.py:58)."""
    if img_list.shape[0] != 1 or coarse.base_channels != flow.base_channels or coarse.out_channels < 32:
        return False
    if not pointflow.conv2d_wide_stacked_supported([coarse.conv0[0].conv, flow.conv0[0].conv]):
        return False
    for tower in (coarse, flow):
        for name in ("conv0", "conv1", "conv2", "conv3"):
            for blk in getattr(tower, name):
                conv, bn = (blk.conv, blk.bn) if hasattr(blk, "bn") else (blk, None)
                if not pointflow.conv2d_wide_preferred(conv):
                    return False
                if bn is not None and not (blk.relu and (bn.training or not bn.track_running_stats)):
                    return False
    return True


def tower_pair_views(coarse, flow, img_list):
    """The coarse and the flow tower of one scene (1,V,3,H,W) side by side: every one of the eleven layers is ONE
    launch -- the first a single 3 -> 8 + 8 convolution of the shared views, the others pf_conv2d_wide_sets_f32
    over 2 V samples (set 0 = coarse tower, set 1 = flow tower; weights, pending BatchNorm and output layout per
    set) -- half the launches and twice the blocks per launch on the small maps (240 -> 480 on 64 x 80).  Per sample the arithmetic is ``forward_views``': bit-identical results.  Returns what
    the fused forward consumes: (coarse "conv3" channel-last (1,V,h,w,C), {"conv1","conv2","conv3"} of the flow
    tower as pointflow.RawLevel)."""
    V = img_list.shape[1]
    x = img_list[0].float().contiguous()                   # (V,3,H,W): the towers' first layer shares its input
    blocks = [(name, cb, fb) for name in ("conv0", "conv1", "conv2", "conv3")
              for cb, fb in zip(getattr(coarse, name), getattr(flow, name))]
    pending = None
    levels = {}
    for i, (name, cb, fb) in enumerate(blocks):
        last = i + 1 == len(blocks)
        stage_end = last or blocks[i + 1][0] != name
        has_bn = hasattr(cb, "bn")
        convs = [cb.conv, fb.conv] if has_bn else [cb, fb]
        if i == 0:       # same input: ONE 3 -> 8 + 8 convolution; its output is read set-interleaved by the next layer
            y, partials = pointflow.conv2d_wide_stacked(x, convs, has_bn)
        else:
            y, partials = pointflow.conv2d_wide_sets(x, convs, pending, 1, has_bn, interleaved=(i == 1),
                                                     channel_last_sets=(0,) if last else ())
        raw_level = stage_end and name != "conv0"          # the flow tower's pyramid level: its rows are needed now
        pending = pointflow.bn_affine_rows_sets(y, [cb.bn, fb.bn], 1, partials, lazy=not raw_level,
                                                interleaved=(i == 0)) if has_bn else None
        if raw_level:
            levels[name] = pointflow.RawLevel(y[V:], None if pending is None else pending.rows_of(1))
        x = y
    C = x.shape[1]
    return x[:V].view(1, V, x.shape[2], x.shape[3], C), levels


def _conv2d_block_fused(block, x, pending, samples_per_stat, defer, lazy=False, channel_last_out=False):
    """One tower block.  ``pending``: BN+ReLU affine rows not yet applied to x.  Returns (y, pending'): with
    ``defer`` the block's own BatchNorm+ReLU is returned as affine rows for the next (custom) conv to apply
    while staging; otherwise y is normalised in place (statistics + fused finalize/normalise)."""
    conv, bn, relu = (block.conv, block.bn, block.relu) if hasattr(block, "bn") else (block, None, False)
    training_bn = bn is not None and (bn.training or not bn.track_running_stats)
    if pointflow.conv2d_wide_preferred(conv):
        y, partials = pointflow.conv2d_wide(x, conv, pending, samples_per_stat, training_bn,
                                            channel_last_out=channel_last_out)
    else:                                      # a shape the tower kernels are not built for: the library convolution
        pointflow.warn_library_fallback("ImageConv layer", conv)
        if pending is not None:
            x = pointflow.channel_affine_(x, pending, True, samples_per_stat)
        y = block._crop(conv(x), x).contiguous() if hasattr(block, "_crop") else conv(x).contiguous()
        partials = None
    if bn is None:
        return (F.relu(y, inplace=True) if relu else y), None
    if defer and relu:
        return y, pointflow.bn_affine_rows(y, bn, samples_per_stat, partials, lazy=lazy)
    return pointflow.batch_norm_act_(y, bn, relu, samples_per_stat, partials=partials), None


class VolumeConv(nn.Module):
    """3-level 3D U-Net regulariser with additive skips; last conv is plain."""

    def __init__(self, in_channels, base_channels):
        super(VolumeConv, self).__init__()
        b = base_channels
        self.in_channels = in_channels
        self.out_channels = 8 * b
        self.base_channels = b
        self.conv1_0 = Conv3d(in_channels, 2 * b, 3, stride=2, padding=1)
        self.conv2_0 = Conv3d(2 * b, 4 * b, 3, stride=2, padding=1)
        self.conv3_0 = Conv3d(4 * b, 8 * b, 3, stride=2, padding=1)
        self.conv0_1 = Conv3d(in_channels, b, 3, 1, padding=1)
        self.conv1_1 = Conv3d(2 * b, 2 * b, 3, 1, padding=1)
        self.conv2_1 = Conv3d(4 * b, 4 * b, 3, 1, padding=1)
        self.conv3_1 = Conv3d(8 * b, 8 * b, 3, 1, padding=1)
        self.conv4_0 = Deconv3d(8 * b, 4 * b, 3, 2, padding=1, output_padding=1)
        self.conv5_0 = Deconv3d(4 * b, 2 * b, 3, 2, padding=1, output_padding=1)
        self.conv6_0 = Deconv3d(2 * b, b, 3, 2, padding=1, output_padding=1)
        self.conv6_2 = nn.Conv3d(b, 1, 3, padding=1, bias=False)

    def _bottom_fusable(self):
        blocks = (self.conv3_0, self.conv3_1, self.conv4_0)
        return (all(b.bn is not None and b.relu for b in blocks)
                and pointflow.conv3d_bottom_supported(self.conv3_0.conv)
                and pointflow.conv3d_bottom_supported(self.conv3_1.conv)
                and pointflow.deconv3d_bottom_supported(self.conv4_0.conv))

    def forward_fused(self, x):
        """Inference fast path: own conv / deconv kernels + HIP BatchNorm/ReLU kernels (statistics pooled over the
        batch); the library convolution only for shapes none of the kernels is built for.  (Variants measured equal
        or slower in round 2 and removed in round 3 -- encoder / decoder BatchNorms resolved by the consuming
        kernels, the skip branches or conv0_1 on streams of their own: DESIGN.md section 6.)"""
        B = x.shape[0]
        f = lambda blk, t: _block_fused(blk, t, B)   # noqa: E731
        blk0, blk6 = self.conv0_1, self.conv6_0
        train0 = blk0.bn is not None and blk0.relu and (blk0.bn.training or not blk0.bn.track_running_stats)
        if train0 and _conv3d_fusable(blk0.conv, x):
            # raw output + statistics: its BatchNorm + ReLU waits for the decoder and rides on the last skip add
            full = pointflow.conv3d_k3(x.contiguous(), blk0.conv.weight, 1, True)
        else:
            full = f(blk0, x)
        pointflow.stamp("conv0_1_end")
        half = f(self.conv1_0, x)
        quarter = f(self.conv2_0, half)
        if self._bottom_fusable():
            # the three smallest layers, one launch each (csrc/conv3d_bottom.hip): every BatchNorm + ReLU between
            # them is applied -- and, with few statistics rows, finalized -- by the NEXT layer while it stages
            b0, b1, b2 = self.conv3_0, self.conv3_1, self.conv4_0
            y0, p0 = pointflow.conv3d_bottom(quarter.contiguous(), b0.conv, None, B, True)
            y1, p1 = pointflow.conv3d_bottom(y0, b1.conv, pointflow.bn_affine_rows(y0, b0.bn, B, p0, lazy=True), B, True)
            pointflow.stamp("unet_encoder_end")
            y2, p2 = pointflow.deconv3d_bottom(y1, b2.conv, pointflow.bn_affine_rows(y1, b1.bn, B, p1, lazy=True), B, True)
            up = pointflow.batch_norm_act_(y2, b2.bn, b2.relu, B, partials=p2)
        else:
            eighth = f(self.conv3_1, f(self.conv3_0, quarter))
            pointflow.stamp("unet_encoder_end")
            up = f(self.conv4_0, eighth)
        pointflow.stamp("unet_bottom_end")
        half = f(self.conv1_1, half)
        quarter = f(self.conv2_1, quarter)
        up = f(self.conv5_0, (up, quarter))
        if (isinstance(full, tuple) and _deconv_fusable(blk6, up) and blk6.bn is not None and blk6.relu
                and (blk6.bn.training or not blk6.bn.track_running_stats)):
            # conv6_0's BatchNorm + ReLU, conv0_1's BatchNorm + ReLU and the add of the two: ONE pass
            y6, p6 = pointflow.deconv3d_k3s2(up.contiguous(), half.contiguous(), blk6.conv.weight, True)
            raw0, p0 = full
            return f(self.conv6_2, pointflow.batch_norm_act2_(raw0, blk0.bn, p0, y6, blk6.bn, p6, B))
        up = f(blk6, (up, half))
        # (conv6_2 does not add on load: its kernel is bound by its tap loads and adding there doubles them --
        # 63 us against 15 + 5, profiles/archive/r01/r01h_microbench_deconv3d.log; the add rides on conv0_1's BatchNorm pass)
        if isinstance(full, tuple):
            raw, partials = full
            summed = pointflow.batch_norm_act_(raw, blk0.bn, blk0.relu, B, partials=partials, addend=up.contiguous())
        else:
            summed = up + full
        return f(self.conv6_2, summed)

    def forward(self, x):
        if pointflow.hip_inference(x, self):        # reference model.py:113-115 without an autograd graph: own kernels
            def run(v):
                y = self.forward_fused(v)
                pointflow.flush_counters()
                return y
            from . import graph
            return graph.module_forward(self, run, x)
        full = self.conv0_1(x)
        half = self.conv1_0(x)
        quarter = self.conv2_0(half)
        eighth = self.conv3_1(self.conv3_0(quarter))
        half = self.conv1_1(half)
        quarter = self.conv2_1(quarter)
        up = self.conv4_0(eighth)
        up = self.conv5_0(up + quarter)
        up = self.conv6_0(up + half)
        return self.conv6_2(up + full)


class MAELoss(nn.Module):
    """Masked mean absolute error in units of the depth interval, summed over the batch
    (reference networks.py:170-181)."""

    def forward(self, pred_depth_image, gt_depth_image, depth_interval):
        interval = depth_interval.view(-1)
        valid = (gt_depth_image != 0.0).float()
        count = valid.sum(dim=(1, 2, 3)) + 1e-7
        err = (valid * (pred_depth_image - gt_depth_image).abs()).sum(dim=(1, 2, 3))
        return ((err / interval) / count).sum()


class Valid_MAELoss(nn.Module):
    """MAELoss restricted to pixels whose previous-stage error is below ``valid_threshold`` intervals
    (reference networks.py:184-207)."""

    def __init__(self, valid_threshold=2.0):
        super(Valid_MAELoss, self).__init__()
        self.valid_threshold = valid_threshold

    def forward(self, pred_depth_image, gt_depth_image, depth_interval, before_depth_image):
        interval = depth_interval.view(-1)
        if before_depth_image.size(2) != pred_depth_image.size(2):
            before_depth_image = F.interpolate(before_depth_image, tuple(pred_depth_image.shape[2:]))
        prior_err = (gt_depth_image - before_depth_image).abs() / interval.view(-1, 1, 1, 1)
        valid = (gt_depth_image != 0.0).float() * (prior_err < self.valid_threshold).float()
        count = valid.sum(dim=(1, 2, 3)) + 1e-7
        err = (valid * (pred_depth_image - gt_depth_image).abs()).sum(dim=(1, 2, 3))
        return ((err / interval) / count).sum()
