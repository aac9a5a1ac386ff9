"""In-degree of the neighbour lists the cfg-4 training step (and the cfg-2 forward) really sees: how many lists name each
point.  The inverted-list gather of the EdgeConv backward walks ONE point's list per lane group, so its time follows the
longest lists of a wave, not the mean of 16."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pointmvsnet_amd import synthetic, train_ops  # noqa: E402
from pointmvsnet_amd.model import PointMVSNet  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
_, _, _, _, _, img_scales, inter_scales = synthetic.CONFIGS[cfg]
data, _, _ = synthetic.make_config(cfg, seed=0, train_intrinsics=True)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
batch["gt_depth_img"] = synthetic.make_gt_depth(data, seed=0).to(dev)
net = PointMVSNet()
synthetic.seed_weights(net, seed=0)
net = net.to(dev).train()
seen = []
orig = train_ops.edge_chain_train


def spy(edge_convs, feature, idx, plane_hw=None):
    seen.append((idx.detach().cpu().numpy(), plane_hw))
    return orig(edge_convs, feature, idx, plane_hw=plane_hw)


train_ops.edge_chain_train = spy
preds = net(batch, img_scales, inter_scales, isFlow=True, isTest=False)
if os.environ.get("WITH_BACKWARD", "0") != "0":          # (under rocprofv3: the eager backward's kernel times)
    from pointmvsnet_amd.model import PointMVSNetLoss
    for _ in range(3):
        losses = PointMVSNetLoss(8.0)(preds, batch, True)
        sum(losses.values()).backward()
        preds = net(batch, img_scales, inter_scales, isFlow=True, isTest=False)
torch.cuda.synchronize()
for idx, hw in seen:
    N = idx.shape[1]
    cnt = np.bincount(idx[0].reshape(-1).clip(0, N - 1), minlength=N)
    rows8 = cnt.reshape(-1, 8).max(axis=1).mean()      # a C=32 wave holds 8 rows of 8 lanes: it lasts as long as its longest list
    print("N %6d plane %s  in-degree mean %.1f  p50 %d  p90 %d  p99 %d  max %d  zeros %.1f %%  mean-of-max-over-8-rows %.1f"
          % (N, hw, cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 90), np.percentile(cnt, 99), cnt.max(),
             100.0 * (cnt == 0).mean(), rows8))
