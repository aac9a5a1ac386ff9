"""FeatureFetcher: the multi-view warp.

Same ``nn.Module`` surface as reference utils/feature_fetcher.py:8-60: forward(feature_maps (B,V,C,H,W),
pts (B,3,N), cam_intrinsics (B,V,3,3), cam_extrinsics (B,V,3,4) or None) -> (B,V,C,N), a fresh
contiguous tensor (the model writes into it in place, model.py:106), differentiable w.r.t. the
feature maps only (the sampling grid is built under no_grad in the reference, :29).
"""
import torch
import torch.nn as nn

from .. import _lib


class _Fetch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, maps, pts, K, E):
        B, V, C, H, W = maps.shape
        N = pts.size(2)
        out = torch.empty((B, V, C, N), dtype=torch.float32, device=maps.device)
        with _lib.on_device(maps.device):
            _lib.call("pf_fetch_forward_f32", _lib.ptr(maps), _lib.ptr(pts), _lib.ptr(K), _lib.ptr(E),
                      _lib.ptr(out), B, V, C, H, W, N, _lib.stream(),
                      algo_bytes=4.0 * B * (V * C * H * W + 3 * N + V * C * N))
        ctx.save_for_backward(pts, K, E if E is not None else torch.empty(0, device=maps.device))
        ctx.has_ext = E is not None
        ctx.shape = (B, V, C, H, W)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        pts, K, E = ctx.saved_tensors
        B, V, C, H, W = ctx.shape
        N = pts.size(2)
        g = grad_out.contiguous()
        grad_maps = torch.empty((B, V, C, H, W), dtype=torch.float32, device=g.device)
        with _lib.on_device(g.device):
            _lib.call("pf_fetch_backward_f32", _lib.ptr(g), _lib.ptr(pts), _lib.ptr(K),
                      _lib.ptr(E) if ctx.has_ext else None, _lib.ptr(grad_maps), B, V, C, H, W, N, _lib.stream(),
                      algo_bytes=4.0 * B * (V * C * H * W + 3 * N + V * C * N))
        return grad_maps, None, None, None


def _prep(feature_maps, pts, cam_intrinsics, cam_extrinsics):
    _lib.require_gpu(feature_maps, pts, cam_intrinsics, cam_extrinsics)
    if feature_maps.dim() != 5 or pts.dim() != 3 or pts.size(1) != 3:
        raise RuntimeError("FeatureFetcher: expected feature_maps (B,V,C,H,W) and pts (B,3,N)")
    B, V = feature_maps.shape[:2]
    f32 = torch.float32
    # (the common case costs nothing: float32, contiguous tensors pass through untouched -- every no-op conversion is a
    # dispatcher round trip, and model.py makes 31 of these calls per depth map)
    maps = feature_maps if (feature_maps.dtype == f32 and feature_maps.is_contiguous()) else feature_maps.float().contiguous()
    p = pts.detach()
    if p.dtype != f32 or not p.is_contiguous():
        p = p.float().contiguous()
    K = cam_intrinsics.detach()
    if K.dtype != f32 or not K.is_contiguous() or K.dim() != 4:
        K = K.float().reshape(B, V, 3, 3).contiguous()
    E = None
    if cam_extrinsics is not None:
        E = cam_extrinsics.detach()
        if E.dtype != f32 or not E.is_contiguous() or E.dim() != 4:
            E = E.float().reshape(B, V, 3, 4).contiguous()
    return maps, p, K, E


class FeatureFetcher(nn.Module):
    def __init__(self, mode="bilinear"):
        super(FeatureFetcher, self).__init__()
        if mode != "bilinear":
            raise NotImplementedError("FeatureFetcher: only the reference default mode 'bilinear' is built")
        self.mode = mode

    def forward(self, feature_maps, pts, cam_intrinsics, cam_extrinsics):
        maps, p, K, E = _prep(feature_maps, pts, cam_intrinsics, cam_extrinsics)
        if not (torch.is_grad_enabled() and maps.requires_grad):
            # nothing to differentiate (the reference's test loop, test.py:61): the launch without an autograd node --
            # model.py makes 31 of these calls per depth map at cfg 2, the node's bookkeeping is most of their host cost
            B, V, C, H, W = maps.shape
            N = p.size(2)
            out = torch.empty((B, V, C, N), dtype=torch.float32, device=maps.device)
            with _lib.on_device(maps.device):
                _lib.call("pf_fetch_forward_f32", _lib.ptr(maps), _lib.ptr(p), _lib.ptr(K), _lib.ptr(E),
                          _lib.ptr(out), B, V, C, H, W, N, _lib.stream(),
                          algo_bytes=4.0 * B * (V * C * H * W + 3 * N + V * C * N))
            return out
        return _Fetch.apply(maps, p, K, E)


def fetch_variance(feature_maps, pts, cam_intrinsics, cam_extrinsics, ref_override=False):
    """Fused rows W+V: variance over views of the fetched features, (B,C,N); inference only.

    ``ref_override`` makes view 0 contribute its un-warped map (reference model.py:103-106)."""
    maps, p, K, E = _prep(feature_maps, pts, cam_intrinsics, cam_extrinsics)
    B, V, C, H, W = maps.shape
    N = p.size(2)
    if ref_override and N % (H * W) != 0:
        raise RuntimeError("fetch_variance: ref_override needs N to be a multiple of H*W")
    out = torch.empty((B, C, N), dtype=torch.float32, device=maps.device)
    with _lib.on_device(maps.device):
        _lib.call("pf_fetch_variance_f32", _lib.ptr(maps), _lib.ptr(p), _lib.ptr(K), _lib.ptr(E), _lib.ptr(out),
                  B, V, C, H, W, N, int(bool(ref_override)), _lib.stream(),
                  algo_bytes=4.0 * B * (V * C * H * W + 3 * N + C * N))
    return out


class ChannelLast(object):
    """Feature maps that already are channel-last: ``maps`` is (B, V, H, W, C) contiguous float32."""

    def __init__(self, maps):
        self.maps = maps


def to_channel_last(maps):
    """(..., C, H, W) contiguous -> (..., H, W, C) contiguous (pf_nchw_to_nhwc_f32)."""
    C, H, W = maps.shape[-3:]
    P = maps.numel() // (C * H * W)
    out = torch.empty(tuple(maps.shape[:-3]) + (H, W, C), dtype=torch.float32, device=maps.device)
    with _lib.on_device(maps.device):
        _lib.call("pf_nchw_to_nhwc_f32", _lib.ptr(maps), _lib.ptr(out), P, C, H * W, _lib.stream(),
                  algo_bytes=8.0 * maps.numel())
    return out


def frustum_variance(feature_maps, kinv, rinv, t, depths, cam_intrinsics, cam_extrinsics, want_points=True,
                     channel_last=None):
    """Coarse cost volume (reference model.py:79-111) without materialising the frustum first: the world
    points of the reference view's depth hypotheses are generated inside the fetch+variance kernel
    (pf_frustum_variance_f32).  kinv/rinv (B,3,3), t (B,3), depths (B,D) float32 on the device.
    Returns (cost (B,C,D*H*W), world_points (B,3,D*H*W) or None)."""
    maps_cl = None
    if isinstance(feature_maps, ChannelLast):                 # produced channel-last by the tower's last layer
        maps_cl = feature_maps.maps.detach().float().contiguous()
        _lib.require_gpu(maps_cl, kinv, rinv, t, depths, cam_intrinsics, cam_extrinsics)
        B, V, H, W, C = maps_cl.shape
        maps, channel_last = maps_cl, True
    else:
        _lib.require_gpu(feature_maps, kinv, rinv, t, depths, cam_intrinsics, cam_extrinsics)
        maps = feature_maps.detach().float().contiguous()
        B, V, C, H, W = maps.shape
    D = depths.shape[-1]
    kinv = kinv.detach().float().reshape(B, 9).contiguous()
    rinv = rinv.detach().float().reshape(B, 9).contiguous()
    t = t.detach().float().reshape(B, 3).contiguous()
    depths = depths.detach().float().reshape(B, D).contiguous()
    K = cam_intrinsics.detach().float().reshape(B, V, 9).contiguous()
    E = cam_extrinsics.detach().float().reshape(B, V, 12).contiguous()
    N = D * H * W
    out = torch.empty((B, C, N), dtype=torch.float32, device=maps.device)
    world = torch.empty((B, 3, N), dtype=torch.float32, device=maps.device) if want_points else None
    if channel_last is None:
        channel_last = C % 4 == 0
    if channel_last:
        if maps_cl is None:
            maps_cl = to_channel_last(maps)
        with _lib.on_device(maps.device):
            _lib.call("pf_frustum_variance_cl_f32", _lib.ptr(maps_cl), _lib.ptr(kinv), _lib.ptr(rinv), _lib.ptr(t),
                      _lib.ptr(depths), _lib.ptr(K), _lib.ptr(E), _lib.ptr(out), _lib.ptr(world), B, V, C, H, W, D,
                      _lib.stream(), algo_bytes=4.0 * B * (V * C * H * W + (3 * N if want_points else 0) + C * N))
        return out, world
    with _lib.on_device(maps.device):
        _lib.call("pf_frustum_variance_f32", _lib.ptr(maps), _lib.ptr(kinv), _lib.ptr(rinv), _lib.ptr(t),
                  _lib.ptr(depths), _lib.ptr(K), _lib.ptr(E), _lib.ptr(out), _lib.ptr(world), B, V, C, H, W, D,
                  _lib.stream(), algo_bytes=4.0 * B * (V * C * H * W + (3 * N if want_points else 0) + C * N))
    return out, world
