"""CPU oracle for the PointFlow hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  Nothing under ``pointmvsnet_amd/`` imports it; the product path fails
loudly when the HIP library is missing instead of falling back to anything in here.

What it is: a functional (state-dict driven) restatement, on torch-CPU float32, of the reference
algorithm for every row of SURVEY.md section 8(a).  Every function cites the reference lines it
follows (paths relative to ``/root/reference``).  The arithmetic of the path lives in PyTorch/ATen
(not vendored in the reference; README.md:33-37 pins "Pytorch 1.0.1" in prose only), so the
restatement calls the same ATen operators the reference calls, with the two documented shims:

* F7: ``grid_sample(..., align_corners=True)`` - the PyTorch-1.0.1 behaviour the reference's grid
  normalisation (utils/feature_fetcher.py:51-53) is written for.
* F6: EdgeConv is evaluated with its CUDA-branch math (neighbours gathered from conv2's output,
  networks.py:26-28), which is the parity target ("match the reference CUDA ops").

Pinning: ``tests/test_oracle_golden.py`` checks this module against golden vectors produced by
running the *unmodified reference code* (plus the same two shims) in the build container
(``tests/golden/make_golden.py``), and against the reference's own two known-answer self tests
(functions/gather_knn.py:27-56, utils/feature_fetcher.py:63-97).  ``oracle/bruteforce.py`` holds
independent first-principles NumPy versions of the irregular ops used to cross-check this file.
"""
import collections

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# row P / U : pixel grid and un-projection
# --------------------------------------------------------------------------------------------
def pixel_grid(height, width):
    """Homogeneous pixel-centre grid [x+.5, y+.5, 1], row-major (functions/functions.py:128-138)."""
    xs = torch.linspace(0.5, width - 0.5, width).view(1, width).expand(height, width)
    ys = torch.linspace(0.5, height - 0.5, height).view(height, 1).expand(height, width)
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width)], dim=0)


def split_cameras(cam_params_list):
    """R, t, R^-1 and raw K from the packed camera tensor (model.py:54-58)."""
    ext = cam_params_list[:, :, 0, :3, :4].clone()
    R = ext[:, :, :3, :3]
    t = ext[:, :, :3, 3].unsqueeze(-1)
    return ext, R, t, torch.inverse(R)


# --------------------------------------------------------------------------------------------
# row W : the warp (utils/feature_fetcher.py:13-60)
# --------------------------------------------------------------------------------------------
def fetch_features(feature_maps, pts, cam_intrinsics, cam_extrinsics):
    """(B,V,C,H,W) maps sampled at the projections of (B,3,N) world points -> (B,V,C,N).

    p = R X + t (feature_fetcher.py:36-40); (x/z, y/z, 1) K^T (:45-49); pixel-index sampling at
    (u-.5, v-.5), bilinear, zero padding, align_corners=True (:51-55 and finding F7).
    """
    B, V, C, H, W = feature_maps.shape
    maps = feature_maps.reshape(B * V, C, H, W)
    K = cam_intrinsics.reshape(B * V, 3, 3)
    N = pts.shape[2]
    with torch.no_grad():
        p = pts.unsqueeze(1).expand(B, V, 3, N).contiguous().view(B * V, 3, N)
        if cam_extrinsics is not None:
            E = cam_extrinsics.reshape(B * V, 3, 4)
            p = torch.bmm(E[:, :, :3], p) + E[:, :, 3:4].expand(B * V, 3, N)
        p = p.to(torch.get_default_dtype()).transpose(1, 2)      # the reference: .float(); float64 only when a test sets it
        x, y, z = p[..., 0], p[..., 1], p[..., 2]
        nuv = torch.stack([x / z, y / z, torch.ones_like(x)], dim=-1)
        uv = torch.bmm(nuv, K.transpose(1, 2))[:, :, :2]
        grid = (uv - 0.5).view(B * V, N, 1, 2)
        grid[..., 0] = (grid[..., 0] / float(W - 1)) * 2 - 1.0
        grid[..., 1] = (grid[..., 1] / float(H - 1)) * 2 - 1.0
    out = F.grid_sample(maps, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.squeeze(3).view(B, V, C, N)


# --------------------------------------------------------------------------------------------
# row V : variance over views (model.py:108-111, :188-190)
# --------------------------------------------------------------------------------------------
def variance_over_views(point_features):
    m1 = torch.mean(point_features, dim=1)
    m2 = torch.mean(point_features ** 2, dim=1)
    return m2 - m1 ** 2


# --------------------------------------------------------------------------------------------
# row K : lattice kNN (utils/torch_utils.py:16-61)
# --------------------------------------------------------------------------------------------
def knn_lattice(xyz, kernel_size=5, knn=16, return_code=False):
    """Centre-minus-candidate differences over the k^3 window via conv3d with +-1 taps and zero
    padding (torch_utils.py:29-44), d2 = sum of squares over xyz (:46-47), topk of -d2 (:49),
    candidate code -> lattice offsets -> linear index with one global clamp (:51-59)."""
    B, _, D, H, W = xyz.shape
    assert kernel_size % 2 == 1
    hk = kernel_size // 2
    k3 = kernel_size ** 3
    w = torch.zeros(3 * k3, 3, kernel_size, kernel_size, kernel_size)
    for axis in range(3):
        for c in range(k3):
            i, j, k = c // (kernel_size ** 2), (c // kernel_size) % kernel_size, c % kernel_size
            w[axis * k3 + c, axis, i, j, k] -= 1.0
            w[axis * k3 + c, axis, hk, hk, hk] += 1.0
    diff = F.conv3d(xyz, w.to(xyz.device), padding=hk).contiguous().view(B, 3, k3, -1)
    d2 = torch.sum(diff ** 2, dim=1)
    _, code = torch.topk(-d2, k=knn, dim=1)
    code = code.permute(0, 2, 1)
    dd = code // (kernel_size ** 2) - hk
    dh = (code % (kernel_size ** 2)) // kernel_size - hk
    dw = code % kernel_size - hk
    base = torch.arange(D * H * W).view(1, -1, 1).expand(B, -1, knn)
    idx = torch.clamp(base + dd * (H * W) + dh * W + dw, 0, D * H * W - 1)
    return (idx, code) if return_code else idx


def knn_lattice_d2(xyz, kernel_size=5):
    """The (B, k^3, N) squared distances the reference ranks (torch_utils.py:44-47); used by the
    parity tests to resolve tie groups (finding F10)."""
    B = xyz.shape[0]
    hk = kernel_size // 2
    k3 = kernel_size ** 3
    w = torch.zeros(3 * k3, 3, kernel_size, kernel_size, kernel_size)
    for axis in range(3):
        for c in range(k3):
            i, j, k = c // (kernel_size ** 2), (c // kernel_size) % kernel_size, c % kernel_size
            w[axis * k3 + c, axis, i, j, k] -= 1.0
            w[axis * k3 + c, axis, hk, hk, hk] += 1.0
    diff = F.conv3d(xyz, w, padding=hk).contiguous().view(B, 3, k3, -1)
    return torch.sum(diff ** 2, dim=1)


# --------------------------------------------------------------------------------------------
# row G : gather_knn (functions/csrc/gather_knn_kernel.cu:25-47 fwd, :50-89 bwd)
# --------------------------------------------------------------------------------------------
def gather_knn(feature, index):
    """out[b,c,n,j] = feature[b,c,index[b,n,j]]; differentiable, so autograd of this expression
    is the oracle for the scatter-add backward kernel (gather_knn_kernel.cu:50-89)."""
    B, C, N = feature.shape
    K = index.shape[2]
    return torch.gather(feature.unsqueeze(2).expand(B, C, N, N), 3,
                        index.unsqueeze(1).expand(B, C, N, K))


def gather_knn_backward(grad_output, index):
    """grad_input[b,c,index[b,n,j]] += grad_output[b,c,n,j] in float64 order-independent form."""
    B, C, N, K = grad_output.shape
    gi = torch.zeros(B, C, N, dtype=grad_output.dtype)
    flat_idx = index.reshape(B, 1, N * K).expand(B, C, N * K)
    gi.scatter_add_(2, flat_idx, grad_output.reshape(B, C, N * K))
    return gi


# --------------------------------------------------------------------------------------------
# rows E0 / E1 / E2 : EdgeConvNoC and EdgeConv (networks.py:9-81), CUDA-branch semantics (F6)
# --------------------------------------------------------------------------------------------
def _batch_norm(x, sd, prefix, track=None):
    """Train-mode BatchNorm with batch statistics (the reference evaluates in model.train(),
    test.py:58).  ``track`` (a dict) receives the updated running statistics when given."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if track is None:
        return F.batch_norm(x, None, None, w, b, True, BN_MOMENTUM, BN_EPS)
    rm = track.setdefault(prefix + ".running_mean", sd[prefix + ".running_mean"].clone())
    rv = track.setdefault(prefix + ".running_var", sd[prefix + ".running_var"].clone())
    track[prefix + ".num_batches_tracked"] = track.get(
        prefix + ".num_batches_tracked", sd[prefix + ".num_batches_tracked"].clone()) + 1
    return F.batch_norm(x, rm, rv, w, b, True, BN_MOMENTUM, BN_EPS)


def edge_conv(feature, knn_inds, sd, prefix, concat, track=None):
    """concat=True : EdgeConv   (networks.py:18-45)  -> (B, 2*C_out, N)
       concat=False: EdgeConvNoC(networks.py:56-81)  -> (B, C_out, N)
    local = conv1(x), edge = conv2(x); neighbours gathered from ``edge`` (CUDA branch,
    networks.py:28/:66); BN over (B, N, k) with batch statistics; ReLU; mean over k."""
    k = knn_inds.shape[2]
    local = F.conv1d(feature, sd[prefix + ".conv1.weight"])
    edge = F.conv1d(feature, sd[prefix + ".conv2.weight"])
    neighbour = gather_knn(edge, knn_inds)
    central = local.unsqueeze(-1).expand(-1, -1, -1, k)
    if concat:
        e = torch.cat([central, neighbour - central], dim=1)
    else:
        e = neighbour - central
    e = _batch_norm(e, sd, prefix + ".bn", track)
    e = F.relu(e)
    return torch.mean(e, dim=3)


# --------------------------------------------------------------------------------------------
# conv stacks: ImageConv (networks.py:84-124), VolumeConv row R (networks.py:127-167),
# SharedMLP row M (nn/mlp.py:45-81, nn/conv.py:7-41)
# --------------------------------------------------------------------------------------------
def _cbr2d(x, sd, prefix, stride, padding, track=None):
    x = F.conv2d(x, sd[prefix + ".conv.weight"], None, stride, padding)
    return F.relu(_batch_norm(x, sd, prefix + ".bn", track))


def image_conv(img, sd, prefix, track=None):
    """Four stages; strides 1/2/2/2, kernels 3,3 | 5,3,3 | 5,3,3 | 5,3,plain-3 (networks.py:89-124)."""
    out = {}
    x = _cbr2d(img, sd, prefix + ".conv0.0", 1, 1, track)
    x = _cbr2d(x, sd, prefix + ".conv0.1", 1, 1, track)
    out["conv0"] = x
    for stage in ("conv1", "conv2"):
        x = _cbr2d(x, sd, "%s.%s.0" % (prefix, stage), 2, 2, track)
        x = _cbr2d(x, sd, "%s.%s.1" % (prefix, stage), 1, 1, track)
        x = _cbr2d(x, sd, "%s.%s.2" % (prefix, stage), 1, 1, track)
        out[stage] = x
    x = _cbr2d(x, sd, prefix + ".conv3.0", 2, 2, track)
    x = _cbr2d(x, sd, prefix + ".conv3.1", 1, 1, track)
    x = F.conv2d(x, sd[prefix + ".conv3.2.weight"], None, 1, 1)
    out["conv3"] = x
    return out


def _cbr3d(x, sd, prefix, stride, track=None):
    x = F.conv3d(x, sd[prefix + ".conv.weight"], None, stride, 1)
    return F.relu(_batch_norm(x, sd, prefix + ".bn", track))


def _dbr3d(x, sd, prefix, track=None):
    x = F.conv_transpose3d(x, sd[prefix + ".conv.weight"], None, 2, 1, 1)
    return F.relu(_batch_norm(x, sd, prefix + ".bn", track))


def volume_conv(x, sd, prefix, track=None):
    """3-level 3D U-Net with additive skips; last conv has no BN/ReLU (networks.py:149-167)."""
    c01 = _cbr3d(x, sd, prefix + ".conv0_1", 1, track)
    c10 = _cbr3d(x, sd, prefix + ".conv1_0", 2, track)
    c20 = _cbr3d(c10, sd, prefix + ".conv2_0", 2, track)
    c30 = _cbr3d(c20, sd, prefix + ".conv3_0", 2, track)
    c11 = _cbr3d(c10, sd, prefix + ".conv1_1", 1, track)
    c21 = _cbr3d(c20, sd, prefix + ".conv2_1", 1, track)
    c31 = _cbr3d(c30, sd, prefix + ".conv3_1", 1, track)
    c40 = _dbr3d(c31, sd, prefix + ".conv4_0", track)
    c50 = _dbr3d(c40 + c21, sd, prefix + ".conv5_0", track)
    c60 = _dbr3d(c50 + c11, sd, prefix + ".conv6_0", track)
    return F.conv3d(c60 + c01, sd[prefix + ".conv6_2.weight"], None, 1, 1)


def flow_mlp(x, sd, prefix, track=None):
    """SharedMLP 224->64->64->16 (conv1d 1x1 + BN1d + ReLU) then Conv1d 16->1 (model.py:40-43)."""
    for i in range(3):
        x = F.conv1d(x, sd["%s.0.%d.conv.weight" % (prefix, i)])
        x = F.relu(_batch_norm(x, sd, "%s.0.%d.bn" % (prefix, i), track))
    return F.conv1d(x, sd[prefix + ".1.weight"])


# --------------------------------------------------------------------------------------------
# row S : soft-argmin and probability map (model.py:117-130, functions/functions.py:141-175)
# --------------------------------------------------------------------------------------------
def soft_argmin(filtered_cost, depth_start, depth_end, num_depth):
    prob = F.softmax(-filtered_cost, dim=1)
    dv = torch.stack([torch.linspace(float(depth_start[i]), float(depth_end[i]), num_depth)
                      for i in range(filtered_cost.shape[0])], dim=0)
    dv = dv.view(-1, num_depth, 1, 1).expand(prob.shape)
    return torch.sum(dv * prob, dim=1).unsqueeze(1), prob


def probability_map(prob_volume, depth_map, depth_start, depth_interval):
    B, _, H, W = depth_map.shape
    D = prob_volume.shape[1]
    idx = ((depth_map - depth_start.view(-1, 1, 1, 1)) / depth_interval.view(-1, 1, 1, 1))
    lo = torch.clamp(idx.floor(), 0, D - 1).long()
    hi = torch.clamp(idx.ceil(), 0, D - 1).long()
    return torch.gather(prob_volume, 1, lo) + torch.gather(prob_volume, 1, hi)


# --------------------------------------------------------------------------------------------
# rows F / T / H and the whole forward (model.py:45-305)
# --------------------------------------------------------------------------------------------
def flow_point_features(pyramids, depth_map, interval, cam_intrinsic, cam_extrinsic, R_inv, t,
                        mean, std):
    """Row F (model.py:165-204): for the five hypotheses depth + i*interval, un-project, fetch the
    three pyramid levels (each first bilinearly resized to the flow grid, model.py:184), take the
    variance over views, append the normalised xyz repeated 8x -> (B,136,5,h*w), (B,3,5,h,w)."""
    B, _, h, w = depth_map.shape
    grid = pixel_grid(h, w).view(1, 1, 3, -1).expand(B, 1, 3, -1)
    uv = torch.matmul(torch.inverse(cam_intrinsic[:, 0]).unsqueeze(1), grid)
    feats, xyzs = [], []
    for i in (-2, -1, 0, 1, 2):
        d = depth_map + interval.view(-1, 1, 1, 1) * i
        cam_pts = uv * d.view(B, 1, 1, -1)
        world = torch.matmul(R_inv[:, 0:1], cam_pts - t[:, 0:1]).transpose(1, 2).contiguous().view(B, 3, -1)
        per_level = []
        for name in ("conv1", "conv2", "conv3"):
            fm = pyramids[name]
            V, c, fh, fw = fm.shape[1:]
            fm = F.interpolate(fm.reshape(-1, c, fh, fw), (h, w), mode="bilinear", align_corners=False)
            fm = fm.view(B, V, c, h, w)
            per_level.append(variance_over_views(fetch_features(fm, world, cam_intrinsic, cam_extrinsic)))
        xyz = (world - mean.unsqueeze(-1)) / std.unsqueeze(-1)
        per_level.append(xyz.repeat(1, 8, 1))
        feats.append(torch.cat(per_level, dim=1))
        xyzs.append(xyz)
    feature = torch.stack(feats, dim=2)
    xyz = torch.stack(xyzs, dim=2).contiguous().view(B, 3, 5, h, w)
    return feature, xyz


def sub_flow(xyz, feature, interval, sd, k, track=None):
    """kNN -> 3 edge convs -> MLP -> softmax over the 5 hypotheses -> expected offset
    (model.py:207-229 / :271-291).  xyz (B,3,5,h,w), feature (B,136,5,h,w)."""
    B, _, D, h, w = xyz.shape
    idx = knn_lattice(xyz, D, knn=k)
    x = feature.contiguous().view(B, -1, D * h * w)
    edges = []
    for li, concat in ((0, False), (1, True), (2, True)):
        x = edge_conv(x, idx, sd, "flow_edge_conv.%d" % li, concat, track)
        edges.append(x)
    flow = flow_mlp(torch.cat(edges, dim=1), sd, "flow_mlp", track).contiguous().view(B, D, h, w)
    prob = F.softmax(-flow, dim=1)
    length = torch.tensor([-2.0, -1.0, 0.0, 1.0, 2.0]).view(1, -1, 1, 1) * interval.view(-1, 1, 1, 1)
    return torch.sum(prob * length, dim=1, keepdim=True), prob


def forward(sd, data_batch, img_scales, inter_scales, is_flow=True, is_test=True, k=16, track=None):
    """PointMVSNet.forward restated over a state dict (model.py:45-305).  Returns the same
    ``preds`` keys: world_points, coarse_depth_map, coarse_prob_map, flow{i}, flow{i}_prob."""
    preds = collections.OrderedDict()
    imgs, cams = data_batch["img_list"], data_batch["cam_params_list"]
    B, V, _, H, W = imgs.shape
    ext, R, t, R_inv = split_cameras(cams)
    K = cams[:, :, 1, :3, :3].clone()
    K[:, :, :2, :3] = K[:, :, :2, :3] / 2.0
    if is_test:
        K[:, :, :2, :3] = K[:, :, :2, :3] / 4.0
    d_start, d_int = cams[:, 0, 1, 3, 0], cams[:, 0, 1, 3, 1]
    D = int(cams[0, 0, 1, 3, 2])
    d_end = d_start + (D - 1) * d_int

    # coarse stage (model.py:71-130)
    maps = [image_conv(imgs[:, v], sd, "coarse_img_conv", track)["conv3"] for v in range(V)]
    fl = torch.stack(maps, dim=1)
    C, FH, FW = fl.shape[2:]
    depths = torch.stack([torch.linspace(float(d_start[b]), float(d_end[b]), D).view(1, 1, D, 1)
                          for b in range(B)], dim=0)
    grid = pixel_grid(FH, FW).view(1, 1, 3, -1).expand(B, 1, 3, -1)
    uv = torch.matmul(torch.inverse(K[:, 0]).unsqueeze(1), grid)
    cam_pts = (uv.unsqueeze(3) * depths).view(B, 1, 3, -1)
    world = torch.matmul(R_inv[:, 0:1], cam_pts - t[:, 0:1]).transpose(1, 2).contiguous().view(B, 3, -1)
    preds["world_points"] = world
    pf = fetch_features(fl, world, K, ext)
    pf[:, 0] = maps[0].unsqueeze(2).expand(-1, -1, D, -1, -1).contiguous().view(B, C, -1)
    cost = variance_over_views(pf).view(B, C, D, FH, FW)
    filtered = volume_conv(cost, sd, "coarse_vol_conv", track).squeeze(1)
    depth_map, prob = soft_argmin(filtered, d_start, d_end, D)
    preds["coarse_depth_map"] = depth_map
    preds["coarse_prob_map"] = probability_map(prob, depth_map, d_start, d_int)
    if not is_flow:
        return preds

    # flow stage (model.py:132-303)
    pyr = {n: [] for n in ("conv1", "conv2", "conv3")}
    for v in range(V):
        o = image_conv(imgs[:, v], sd, "flow_img_conv", track)
        for n in pyr:
            pyr[n].append(o[n])
    pyr = {n: torch.stack(vs, dim=1) for n, vs in pyr.items()}
    if is_test:
        pyr = {n: v.detach() for n, v in pyr.items()}

    cur = depth_map
    for it, (s, inter) in enumerate(zip(img_scales, inter_scales)):
        if is_test:
            cur = cur.detach()
        interval = inter * d_int
        h, w = int(H * s), int(W * s)
        if cur.shape[2] != h:
            cur = F.interpolate(cur, (h, w), mode="nearest")
        Kf = cams[:, :, 1, :3, :3].clone()
        Kf[:, :, :2, :3] *= s if is_test else 4 * s
        feature, xyz = flow_point_features(pyr, cur, interval, Kf, ext, R_inv, t,
                                           data_batch["mean"], data_batch["std"])
        if (not is_test) or s == 0.125:
            flow, fprob = sub_flow(xyz, feature, interval, sd, k, track)
        elif s in (0.25, 0.5, 1.0):
            r = int(s * 8)                      # row T: r*r strided sub-lattices, sequential
            hs, ws = h // r, w // r
            f7 = feature.view(B, -1, 5, hs, r, ws, r)
            x7 = xyz.view(B, 3, 5, hs, r, ws, r)
            flow = torch.zeros(B, 1, hs, r, ws, r)
            fprob = torch.zeros(B, 5, hs, r, ws, r)
            for i in range(r):
                for j in range(r):
                    fij, pij = sub_flow(x7[:, :, :, :, i, :, j], f7[:, :, :, :, i, :, j], interval, sd, k, track)
                    flow[:, :, :, i, :, j] = fij
                    fprob[:, :, :, i, :, j] = pij
            flow = flow.view(B, 1, h, w)
            fprob = fprob.view(B, 5, h, w)
        else:
            raise NotImplementedError
        preds["flow%d_prob" % (it + 1)] = fprob
        cur = cur + flow
        preds["flow%d" % (it + 1)] = cur
    return preds
