"""One data-parallel training step of BASELINE config 4 (reference train.py:46-112, solver.py:17-52).

The reference's loop body is  preds = model(batch); optimizer.zero_grad(); loss = sum(loss_fn(...));
loss.backward(); optimizer.step()  under nn.DataParallel with RMSprop(lr 1e-3, alpha 0.9), weight decay off the
``.bn.`` parameters.  Here one process drives one GPU with one scene; replicas exchange gradients with ONE
in-place SUM all-reduce of the flat bucket every parameter's ``.grad`` is a view of (distributed.GradBucket).
The PointFlow stage differentiates through the fused EdgeConv node (networks._EdgeConvTrain: recompute backward,
no (B,2C,N,k) tensor), the HIP FeatureFetcher forward/backward and stock ATen for the rest.
"""
import torch

from . import distributed
from .model import PointMVSNetLoss


def param_groups(module, weight_decay):
    """Decay everything except the BatchNorm parameters (reference solver.py:33-52: names containing '.bn.')."""
    decay, no_decay = [], []
    for name, p in module.named_parameters():
        (no_decay if ".bn." in name else decay).append(p)
    return [dict(params=decay, weight_decay=weight_decay), dict(params=no_decay, weight_decay=0.0)]


class TrainStep(object):
    def __init__(self, model, valid_threshold=8.0, lr=1e-3, alpha=0.9, weight_decay=0.0, group=None):
        self.model = model
        self.loss_fn = PointMVSNetLoss(valid_threshold)            # reference config.py MODEL.VALID_THRESHOLD
        self.bucket = distributed.GradBucket(model)
        self.optimizer = torch.optim.RMSprop(param_groups(model, weight_decay), lr=lr, alpha=alpha)
        self.group = group

    def __call__(self, batch, img_scales, inter_scales, is_flow=True):
        """batch: the reference's data_batch (img_list, cam_params_list, mean, std, gt_depth_img) on the device.
        Returns (total loss (detached), loss dict, preds)."""
        self.model.train()
        self.bucket.zero_()                                        # optimizer.zero_grad(), keeping the views
        preds = self.model(batch, img_scales, inter_scales, isFlow=is_flow, isTest=False)
        losses = self.loss_fn(preds, batch, is_flow)
        total = sum(losses.values())
        total.backward()
        self.bucket.allreduce_sum(self.group)                      # SUM: the loss sums over the batch (networks.py:176-179)
        self.optimizer.step()
        return total.detach(), losses, preds
