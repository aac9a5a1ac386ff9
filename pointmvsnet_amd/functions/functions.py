"""Small helpers of the operator layer plus the DGCNN helpers the
reference package exports but its model never calls (reference functions/functions.py:9-125).

``get_pixel_grids`` / ``get_propability_map`` keep the reference signatures
(functions/functions.py:128-138, :141-175).  The fused pipeline computes both inside HIP kernels
(pixel centres in ``pf_flow_features_f32``, the probability map in ``pf_softargmin_prob_f32``); these
torch versions exist so reference-style callers keep working on either device.
"""
import torch

from .gather_knn import gather_knn


# Set by compat.install_as_pointmvsnet() (the drop-in route for the reference's model.py): get_pixel_grids then returns
# its grid ON THE CURRENT GPU, built once per (height, width, device).  model.py:87-88,166-167 calls it once per stage and
# moves the result with ``.to(img_list.device)`` -- from pageable host memory that is a SYNCHRONOUS copy of up to 1 MB
# which drains the stream three times per depth map (profiles/r06b_route_profile.md: aten::copy_ = 44 % of the route's
# host time); on the device ``.to()`` is a no-op.  Off by default: the reference's function returns a host tensor and a
# direct caller may rely on that.
PIXEL_GRID_ON_DEVICE = False
_GRID_CACHE = {}


def get_pixel_grids(height, width):
    """(3, height*width): rows x+0.5, y+0.5, 1 in row-major pixel order."""
    if PIXEL_GRID_ON_DEVICE and torch.cuda.is_available():
        key = (int(height), int(width), torch.cuda.current_device())
        grid = _GRID_CACHE.get(key)
        if grid is None:
            if len(_GRID_CACHE) >= 16:
                _GRID_CACHE.clear()
            grid = _GRID_CACHE[key] = _pixel_grids_host(height, width).to(torch.device("cuda", key[2]))
        return grid.clone()                 # a fresh tensor per call, like the reference's
    return _pixel_grids_host(height, width)


def _pixel_grids_host(height, width):
    with torch.no_grad():
        cols = torch.arange(width, dtype=torch.float32)
        rows = torch.arange(height, dtype=torch.float32)
        # linspace(0.5, n-0.5, n) of the reference is exactly i + 0.5 for these sizes in float32
        xs = torch.linspace(0.5, width - 0.5, width) if width > 1 else cols + 0.5
        ys = torch.linspace(0.5, height - 0.5, height) if height > 1 else rows + 0.5
        gx = xs.view(1, width).expand(height, width).reshape(-1)
        gy = ys.view(height, 1).expand(height, width).reshape(-1)
        return torch.stack([gx, gy, torch.ones(height * width)], dim=0)


def get_propability_map(cv, depth_map, depth_start, depth_interval):
    """Sum of the two probability-volume entries that bracket the regressed depth.

    cv (B,D,H,W) softmax volume, depth_map (B,1,H,W) -> (B,1,H,W).  Index = (depth-start)/interval,
    floor and ceil each clamped to [0, D-1]."""
    with torch.no_grad():
        D = cv.size(1)
        pos = (depth_map - depth_start.view(-1, 1, 1, 1)) / depth_interval.view(-1, 1, 1, 1)
        pos = pos.detach()
        lower = pos.floor().clamp(0, D - 1).long()
        upper = pos.ceil().clamp(0, D - 1).long()
        return cv.gather(1, lower) + cv.gather(1, upper)


# ---- generic DGCNN helpers (package surface only; model.py does not use them) ----------------------
def pdist(feature):
    """Squared pairwise distance (B,N,N) of (B,C,N) features (reference nn/functional.py:9-26)."""
    sq = (feature * feature).sum(dim=1, keepdim=True)
    inner = torch.bmm(feature.transpose(1, 2), feature)
    return sq.transpose(1, 2) + sq - 2.0 * inner


def get_knn_inds(pdist_mat, k=20, remove=False):
    """k smallest entries per row of a (B,N,N) distance matrix; ``remove`` drops the first pick."""
    take = k + 1 if remove else k
    inds = torch.topk(pdist_mat, take, largest=False, sorted=False)[1]
    return inds[..., 1:] if remove else inds


def construct_edge_feature(feature, knn_inds):
    """(B, 2C, N, k) = cat(central, neighbour - central) using the HIP gather."""
    k = knn_inds.size(-1)
    central = feature.unsqueeze(3).expand(-1, -1, -1, k)
    neighbour = gather_knn(feature, knn_inds)
    return torch.cat((central, neighbour - central), 1)


construct_edge_feature_gather = construct_edge_feature
construct_edge_feature_index = construct_edge_feature


def get_edge_feature(feature, k):
    with torch.no_grad():
        knn_inds = get_knn_inds(pdist(feature), k)
    return construct_edge_feature(feature, knn_inds)
