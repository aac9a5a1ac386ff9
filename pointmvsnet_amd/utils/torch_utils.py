"""Lattice kNN operator and seeding helper.

``get_knn_3d`` keeps the reference signature (utils/torch_utils.py:16-22): xyz (B,3,D,H,W) -> int64
(B, D*H*W, knn).  It accepts the non-contiguous strided sub-lattice views the model builds
(reference model.py:251-252) without a copy: the five strides are handed to the HIP kernel.
"""
import ctypes
import os
import random

import numpy as np
import torch

from .. import _lib


def set_random_seed(seed):
    if seed < 0:
        return
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def knn_lattice(xyz, kernel_size=5, knn=16, with_codes=False, with_idx=True):
    """HIP lattice kNN.  Returns idx, or (idx, codes uint8) when ``with_codes`` (idx is None when
    ``with_idx`` is False: the fused EdgeConv passes consume the 16-byte window codes directly)."""
    _lib.require_gpu(xyz)
    if xyz.dim() != 5 or xyz.size(1) != 3:
        raise RuntimeError("get_knn_3d: xyz must be (B,3,D,H,W)")
    assert kernel_size % 2 == 1            # reference torch_utils.py:24
    if xyz.dtype != torch.float32:
        xyz = xyz.float()
    B, _, D, H, W = xyz.shape
    if knn > kernel_size ** 3:
        raise RuntimeError("get_knn_3d: knn larger than the window (topk would be out of range)")
    if not (with_idx or with_codes):
        raise RuntimeError("knn_lattice: nothing to compute")
    # (the insertion-list kernels -- windows / k the sorting-network kernel is not built for -- always write the
    # int64 indices: give them a buffer even when only the codes are wanted)
    need_idx = with_idx or knn > 16 or kernel_size not in (3, 5)
    idx = torch.empty((B, D * H * W, knn), dtype=torch.int64, device=xyz.device) if need_idx else None
    codes = torch.empty((B, D * H * W, knn), dtype=torch.uint8, device=xyz.device) if with_codes else None
    strides = (ctypes.c_int64 * 5)(*xyz.stride())
    with _lib.on_device(xyz.device):
        _lib.call("pf_knn_lattice_f32", _lib.ptr(xyz), strides, B, D, H, W, int(kernel_size), int(knn),
                  _lib.ptr(idx), _lib.ptr(codes), _lib.stream(),
                  algo_bytes=float(B * D * H * W) * (12.0 + (8.0 * knn if with_idx else 0.0) + (knn if with_codes else 0.0)))
    if not with_idx:
        idx = None
    return (idx, codes) if with_codes else idx


def get_knn_3d(xyz, kernel_size=5, knn=20):
    """k nearest neighbours inside the kernel_size^3 lattice window around every point.

    Distances rank as in the reference (float32 (dx^2+dy^2)+dz^2, zero padding outside the lattice);
    ties resolve to the smaller window code (the reference leaves tie order unspecified)."""
    return knn_lattice(xyz, kernel_size, knn, with_codes=False)
