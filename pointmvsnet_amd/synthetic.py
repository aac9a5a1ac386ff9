"""Seeded synthetic DTU-like inputs and deterministic weights.

There is no DTU data and no checkpoint in the build or bench containers, so every
measurement and every parity test runs on the inputs made here.  The same functions
are used by ``tests/golden/make_golden.py`` (which drives the *reference* code in the
build container), by the parity tests, by ``bench.py`` and by ``__graft_entry__.smoke``,
so "identical inputs" means bit-identical tensors.

Tensor layout follows the reference data pipeline (reference ``pointmvsnet/dataset.py:238-307``,
``pointmvsnet/utils/io.py:15-52``):

* ``img_list``          (B, V, 3, H, W) float32, per-image standardised in the real pipeline
* ``cam_params_list``   (B, V, 2, 4, 4): ``[:, v, 0]`` 4x4 extrinsic [R|t] (p_cam = R p_world + t),
  ``[:, v, 1, :3, :3]`` intrinsic K, ``[:, v, 1, 3, :3]`` = (depth_start, depth_interval, num_depth)
* ``mean`` / ``std``    (B, 3) world-point normalisation constants (reference ``dataset.py:26-27``)
"""
import math

import torch

# reference pointmvsnet/dataset.py:26-27
DTU_MEAN = (1.97145182, -1.52387525, 651.07223895)
DTU_STD = (84.45612252, 93.22252387, 80.08551226)

# Named workloads == BASELINE.json "configs" (SURVEY.md section 8 (d)); heights/widths are the
# legal (multiple-of-64) interpretations documented there.
CONFIGS = {
    # name: (H, W, V, D, inter_scale, img_scales, inter_scales)
    "cfg1": (512, 640, 3, 48, 4.24, (0.125,), (1.0,)),
    "cfg2": (512, 640, 3, 48, 4.24, (0.125, 0.25), (1.0, 0.75)),
    "cfg3": (960, 1280, 5, 96, 2.13, (0.125, 0.25, 0.5), (1.0, 0.75, 0.15)),
    "cfg5": (1152, 1600, 7, 96, 2.13, (0.125, 0.25, 0.5), (1.0, 0.75, 0.15)),
    # BASELINE config 4: the TRAINING step (one scene per GPU; reference config.py:61-63 train scales)
    "cfg4": (512, 640, 3, 48, 4.24, (0.125, 0.25), (0.75, 0.375)),
    # small legal shapes for fast parity tests (not BASELINE configs)
    "tiny": (128, 192, 3, 8, 4.24, (0.125, 0.25), (1.0, 0.75)),
    "small": (256, 320, 3, 16, 4.24, (0.125, 0.25, 0.5), (1.0, 0.75, 0.15)),
    # BASELINE config 5's shape class at a size the CPU reference finishes in seconds: 7 views, 3 iterations
    # including the 16-sub-grid scale 0.5 (golden: tests/golden/model_cfg5r_test.npz)
    "cfg5r": (256, 320, 7, 16, 2.13, (0.125, 0.25, 0.5), (1.0, 0.75, 0.15)),
}


def _rotation(axis, angle):
    """Rodrigues formula in float64 (host side, deterministic)."""
    axis = axis / axis.norm()
    x, y, z = axis.tolist()
    c, s = math.cos(angle), math.sin(angle)
    C = 1.0 - c
    return torch.tensor([
        [c + x * x * C, x * y * C - z * s, x * z * C + y * s],
        [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
        [z * x * C - y * s, z * y * C + x * s, c + z * z * C],
    ], dtype=torch.float64)


def make_scene(height, width, num_view, num_depth, inter_scale=4.24, seed=0, batch=1,
               train_intrinsics=False):
    """Build one synthetic batch on the CPU (float32).

    Intrinsics follow DTU proportions: a 160x128 depth grid has fx~361.5, fy~360.4,
    cx~82.9, cy~66.4; test mode carries full-resolution K (the model divides by 8,
    reference model.py:59-61), train mode carries K of the H/4 grid (model.py:59, :162-163).
    """
    g = torch.Generator().manual_seed(int(seed))
    img = torch.randn(batch, num_view, 3, height, width, generator=g, dtype=torch.float32)
    cams = torch.zeros(batch, num_view, 2, 4, 4, dtype=torch.float64)
    s = width / 160.0
    if train_intrinsics:
        s = s / 4.0
    for b in range(batch):
        for v in range(num_view):
            axis = torch.randn(3, generator=g, dtype=torch.float64)
            angle = 0.02 + 0.03 * v + 0.01 * float(torch.rand(1, generator=g, dtype=torch.float64))
            R = _rotation(axis, angle)
            centre = torch.tensor([30.0 * v + 3.0, -4.0 * v + 2.0, 1.5 * v - 5.0], dtype=torch.float64)
            centre = centre + torch.randn(3, generator=g, dtype=torch.float64)
            t = -R @ centre
            cams[b, v, 0, :3, :3] = R
            cams[b, v, 0, :3, 3] = t
            cams[b, v, 0, 3, 3] = 1.0
            jitter = 1.0 + 0.002 * float(torch.randn(1, generator=g, dtype=torch.float64))
            cams[b, v, 1, 0, 0] = 361.5 * s * jitter
            cams[b, v, 1, 1, 1] = 360.4 * s * jitter
            cams[b, v, 1, 0, 2] = 82.9 * s
            cams[b, v, 1, 1, 2] = 66.4 * s
            cams[b, v, 1, 2, 2] = 1.0
            cams[b, v, 1, 3, 0] = 425.0
            cams[b, v, 1, 3, 1] = 2.5 * inter_scale
            cams[b, v, 1, 3, 2] = float(num_depth)
            cams[b, v, 1, 3, 3] = 425.0 + 2.5 * inter_scale * (num_depth - 1)
    mean = torch.tensor(DTU_MEAN, dtype=torch.float32).view(1, 3).repeat(batch, 1)
    std = torch.tensor(DTU_STD, dtype=torch.float32).view(1, 3).repeat(batch, 1)
    return {
        "img_list": img,
        "cam_params_list": cams.to(torch.float32),
        "mean": mean,
        "std": std,
    }


def make_gt_depth(data, scale=0.25, seed=0):
    """Synthetic ground-truth depth (B, 1, H*scale, W*scale) for the training step (reference dataset.py:285-300
    delivers it at the resolution of the finest flow stage): a smooth surface inside the hypothesis range with a
    band of invalid (zero) pixels, which the masked losses must ignore (reference networks.py:170-207)."""
    img, cams = data["img_list"], data["cam_params_list"]
    B, _, _, H, W = img.shape
    h, w = int(H * scale), int(W * scale)
    g = torch.Generator().manual_seed(1000 + int(seed))
    ys = torch.linspace(0.0, 1.0, h).view(1, 1, h, 1)
    xs = torch.linspace(0.0, 1.0, w).view(1, 1, 1, w)
    gt = torch.zeros(B, 1, h, w)
    for b in range(B):
        start, interval, D = float(cams[b, 0, 1, 3, 0]), float(cams[b, 0, 1, 3, 1]), float(cams[b, 0, 1, 3, 2])
        a, c = torch.rand(2, generator=g).tolist()
        surf = 0.3 + 0.25 * torch.sin(3.0 * xs + 6.0 * a) * torch.cos(2.0 * ys + 6.0 * c) + 0.15 * ys
        gt[b] = start + interval * (D - 1) * surf[0]
    gt[:, :, : max(1, h // 10)] = 0.0
    return gt


def make_config(name, seed=0, batch=1, train_intrinsics=False):
    h, w, v, d, inter, img_scales, inter_scales = CONFIGS[name]
    data = make_scene(h, w, v, d, inter_scale=inter, seed=seed, batch=batch,
                      train_intrinsics=train_intrinsics)
    return data, img_scales, inter_scales


def seed_weights(module, seed=0):
    """Overwrite every parameter/buffer of ``module`` with values that depend only on
    (seed, key name, shape) so that two independently written models with the same
    state-dict keys (the reference's and ours) carry bit-identical weights without
    shipping a checkpoint.  Conv/linear weights get a xavier-like scale, BN affine
    parameters are perturbed around (1, 0), running stats around (0, 1).
    """
    import zlib
    sd = module.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            ref = sd[key]
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * int(seed)) % (2 ** 31))
            if key.endswith("num_batches_tracked"):
                ref.zero_()
                continue
            if key.endswith("running_mean"):
                val = 0.05 * torch.randn(ref.shape, generator=g)
            elif key.endswith("running_var"):
                val = 1.0 + 0.1 * torch.rand(ref.shape, generator=g)
            elif ref.dim() == 1 and key.endswith("weight"):      # BN gamma
                val = 1.0 + 0.1 * torch.randn(ref.shape, generator=g)
            elif ref.dim() == 1:                                  # BN beta / conv bias
                val = 0.1 * torch.randn(ref.shape, generator=g)
            else:                                                 # conv kernels
                fan_out = ref.shape[0] * int(torch.tensor(ref.shape[2:]).prod()) if ref.dim() > 2 else ref.shape[0]
                fan_in = ref.shape[1] * int(torch.tensor(ref.shape[2:]).prod()) if ref.dim() > 2 else ref.shape[1]
                bound = math.sqrt(6.0 / float(fan_in + fan_out))
                val = (torch.rand(ref.shape, generator=g) * 2.0 - 1.0) * bound
            ref.copy_(val.to(ref.dtype))
    return module
