"""First-principles NumPy versions of the irregular ops  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

These do not call any ATen operator: explicit loops / index arithmetic in float64 or float32, small
sizes only.  They cross-check ``oracle/pointflow_oracle.py`` (which calls the same ATen ops the
reference calls) so the oracle is validated from two independent directions: against the reference
run here (tests/golden) and against the written-down math (this file).

Only ``tests/`` may import this module.
"""
import numpy as np


def fetch_bilinear(maps, pts, K, E):
    """Row W from the definition (reference utils/feature_fetcher.py:13-60 + finding F7):
    sample maps[b,v,:, y, x] at pixel index (u-.5, v-.5), bilinear, zeros outside.
    maps (B,V,C,H,W), pts (B,3,N), K (B,V,3,3), E (B,V,3,4) or None -> (B,V,C,N) float64."""
    maps = np.asarray(maps, np.float64)
    B, V, C, H, W = maps.shape
    N = pts.shape[2]
    out = np.zeros((B, V, C, N))
    for b in range(B):
        for v in range(V):
            X = np.asarray(pts[b], np.float64)
            if E is not None:
                X = np.asarray(E[b, v, :, :3], np.float64) @ X + np.asarray(E[b, v, :, 3:4], np.float64)
            nx, ny = X[0] / X[2], X[1] / X[2]
            Kv = np.asarray(K[b, v], np.float64)
            u = Kv[0, 0] * nx + Kv[0, 1] * ny + Kv[0, 2]
            w_ = Kv[1, 0] * nx + Kv[1, 1] * ny + Kv[1, 2]
            ix, iy = u - 0.5, w_ - 0.5
            x0, y0 = np.floor(ix), np.floor(iy)
            for dy in (0, 1):
                for dx in (0, 1):
                    xx, yy = x0 + dx, y0 + dy
                    wgt = (1 - np.abs(ix - xx)) * (1 - np.abs(iy - yy))
                    ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1) & np.isfinite(ix) & np.isfinite(iy)
                    xi = np.where(ok, xx, 0).astype(np.int64)
                    yi = np.where(ok, yy, 0).astype(np.int64)
                    out[b, v] += maps[b, v][:, yi, xi] * (wgt * ok)[None, :]
    return out


def knn_window_d2(xyz, kernel_size=5):
    """Squared distances to the k^3 window candidates in float32 with the reference's rounding:
    each difference is centre - candidate (candidate = 0 outside the lattice, torch_utils.py:44),
    d2 = (dx*dx + dy*dy) + dz*dz (torch_utils.py:46-47).  xyz (3,D,H,W) -> (k^3, D*H*W) float32."""
    xyz = np.asarray(xyz, np.float32)
    _, D, H, W = xyz.shape
    hk = kernel_size // 2
    pad = np.zeros((3, D + 2 * hk, H + 2 * hk, W + 2 * hk), np.float32)
    pad[:, hk:hk + D, hk:hk + H, hk:hk + W] = xyz
    out = np.zeros((kernel_size ** 3, D, H, W), np.float32)
    c = 0
    for i in range(kernel_size):
        for j in range(kernel_size):
            for k in range(kernel_size):
                diff = xyz - pad[:, i:i + D, j:j + H, k:k + W]
                sq = (diff * diff).astype(np.float32)
                out[c] = ((sq[0] + sq[1]).astype(np.float32) + sq[2]).astype(np.float32)
                c += 1
    return out.reshape(kernel_size ** 3, -1)


def knn_window(xyz, kernel_size=5, knn=16):
    """Row K with the stated tie rule of the HIP kernel: smaller d2 first, then smaller candidate
    code.  Returns (idx (N,knn) int64, code (N,knn)).  Index = n + offsets, one global clamp
    (torch_utils.py:51-59)."""
    _, D, H, W = np.asarray(xyz).shape
    hk = kernel_size // 2
    d2 = knn_window_d2(xyz, kernel_size)                      # (k3, N)
    k3, N = d2.shape
    order = np.lexsort((np.broadcast_to(np.arange(k3)[:, None], d2.shape), d2), axis=0)[:knn]  # (knn,N)
    code = order.T
    dd = code // (kernel_size ** 2) - hk
    dh = (code % (kernel_size ** 2)) // kernel_size - hk
    dw = code % kernel_size - hk
    idx = np.arange(N)[:, None] + dd * (H * W) + dh * W + dw
    return np.clip(idx, 0, D * H * W - 1).astype(np.int64), code


def gather(feature, index):
    """Row G forward: out[b,c,n,j] = feature[b,c,index[b,n,j]]."""
    B, C, N = feature.shape
    out = np.zeros((B, C, N, index.shape[2]), feature.dtype)
    for b in range(B):
        out[b] = feature[b][:, index[b]]
    return out


def scatter_add(grad_output, index):
    """Row G backward in float64."""
    B, C, N, K = grad_output.shape
    gi = np.zeros((B, C, N), np.float64)
    for b in range(B):
        for n in range(N):
            for j in range(K):
                gi[b, :, index[b, n, j]] += grad_output[b, :, n, j]
    return gi


def edge_conv(x, idx, w1, w2, gamma, beta, concat, eps=1e-5):
    """Rows E0-E2 in float64 from the definition (networks.py:18-45 / :56-81, CUDA branch):
    y = mean_k relu(BN(cat[l, e[idx]-l])) with batch statistics over (B, N, k), biased variance."""
    x = np.asarray(x, np.float64)
    l = np.einsum("oc,bcn->bon", np.asarray(w1, np.float64), x)
    e = np.einsum("oc,bcn->bon", np.asarray(w2, np.float64), x)
    nb = gather(e, idx)
    cen = np.repeat(l[..., None], idx.shape[2], axis=3)
    t = np.concatenate([cen, nb - cen], axis=1) if concat else nb - cen
    mean = t.mean(axis=(0, 2, 3), keepdims=True)
    var = t.var(axis=(0, 2, 3), keepdims=True)
    t = (t - mean) / np.sqrt(var + eps) * np.asarray(gamma, np.float64).reshape(1, -1, 1, 1) \
        + np.asarray(beta, np.float64).reshape(1, -1, 1, 1)
    n_el = t.shape[0] * t.shape[2] * t.shape[3]
    stats = (mean.reshape(-1), var.reshape(-1) * n_el / (n_el - 1.0))
    return np.maximum(t, 0).mean(axis=3), stats
