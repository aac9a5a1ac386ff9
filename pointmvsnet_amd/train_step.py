"""One data-parallel training step of BASELINE config 4 (reference train.py:46-112, solver.py:17-52).

The reference's loop body is  preds = model(batch); optimizer.zero_grad(); loss = sum(loss_fn(...));
loss.backward(); optimizer.step()  under nn.DataParallel with RMSprop(lr 1e-3, alpha 0.9), weight decay off the
``.bn.`` parameters.  Here one process drives one GPU with one scene; replicas exchange gradients with ONE
in-place SUM all-reduce of the flat bucket every parameter's ``.grad`` is a view of (distributed.GradBucket).
The PointFlow stage differentiates through the fused EdgeConv node (networks._EdgeConvTrain: recompute backward,
no (B,2C,N,k) tensor), the HIP FeatureFetcher forward/backward and stock ATen for the rest.
"""
import torch

from . import distributed, pointflow, train_ops
from .model import PointMVSNetLoss, join_fork_streams


def param_groups(module, weight_decay):
    """Decay everything except the BatchNorm parameters (reference solver.py:33-52: names containing '.bn.')."""
    decay, no_decay = [], []
    for name, p in module.named_parameters():
        (no_decay if ".bn." in name else decay).append(p)
    return [dict(params=decay, weight_decay=weight_decay), dict(params=no_decay, weight_decay=0.0)]


class FlatRMSprop(object):
    """torch.optim.RMSprop(param_groups(model, weight_decay), lr, alpha) (reference solver.py:17-52) as ONE launch per
    step (pf_rmsprop_f32): the parameters move into one flat float32 buffer -- every ``p.data`` becomes a view of it, in
    the bucket's order, so parameter i's gradient sits at the same offset of ``bucket.flat`` -- beside a flat
    square-average buffer.  PyTorch's foreach form is five multi-tensor launches over 115 tensors plus their host-side
    grouping, the only per-step work outside the captured graph.  ``step()`` updates in place; ``state_dict()`` /
    ``load_state_dict()`` carry the square averages per parameter index like torch's optimizer state."""

    def __init__(self, bucket, named_parameters, lr=1e-3, alpha=0.9, eps=1e-8, weight_decay=0.0):
        self.bucket, self.lr, self.alpha, self.eps = bucket, float(lr), float(alpha), float(eps)
        params = bucket.params
        dev = bucket.flat.device
        self.flat = torch.empty_like(bucket.flat)
        self.square_avg = torch.zeros_like(bucket.flat)
        names = {id(p): n for n, p in named_parameters}
        self.wd = None
        if weight_decay != 0.0:
            self.wd = torch.zeros_like(bucket.flat)
        offset = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[offset:offset + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[offset:offset + n].view(p.shape)          # (packed-weight caches re-pack: new storage)
                if self.wd is not None and ".bn." not in names.get(id(p), ""):
                    self.wd[offset:offset + n] = weight_decay
                offset += n
        self.device = dev

    def attached(self):
        base = self.flat.untyped_storage().data_ptr()
        return all(p.data.untyped_storage().data_ptr() == base for p in self.bucket.params)

    def step(self):
        if not self.attached():
            raise RuntimeError("FlatRMSprop: a parameter's storage was replaced (module.to() / load on a new tensor?); "
                               "build a new TrainStep")
        from . import _lib
        with torch.cuda.device(self.device):
            _lib.call("pf_rmsprop_f32", _lib.ptr(self.flat), _lib.ptr(self.bucket.flat), _lib.ptr(self.square_avg),
                      _lib.ptr(self.wd), self.flat.numel(), self.lr, self.alpha, self.eps, _lib.stream(),
                      algo_bytes=20.0 * self.flat.numel())
        # torch's optimizers update parameters through in-place ops, which the packed-weight caches watch through the
        # version counter; the flat kernel writes behind autograd's back, so tick the counters
        for p in self.bucket.params:
            torch.autograd.graph.increment_version(p)

    def state_dict(self):
        out, offset = {}, 0
        for i, p in enumerate(self.bucket.params):
            n = p.numel()
            out[i] = {"square_avg": self.square_avg[offset:offset + n].view(p.shape).clone()}
            offset += n
        return {"state": out, "lr": self.lr, "alpha": self.alpha, "eps": self.eps}

    def load_state_dict(self, sd):
        offset = 0
        for i, p in enumerate(self.bucket.params):
            n = p.numel()
            if i in sd["state"]:
                self.square_avg[offset:offset + n].copy_(sd["state"][i]["square_avg"].reshape(-1))
            offset += n


class TrainStep(object):
    def __init__(self, model, valid_threshold=8.0, lr=1e-3, alpha=0.9, weight_decay=0.0, group=None):
        self.model = model
        self.loss_fn = PointMVSNetLoss(valid_threshold)            # reference config.py MODEL.VALID_THRESHOLD
        self.bucket = distributed.GradBucket(model)
        if self.bucket.flat.is_cuda:
            self.optimizer = FlatRMSprop(self.bucket, list(model.named_parameters()), lr=lr, alpha=alpha,
                                         weight_decay=weight_decay)
        else:                                                      # (CPU: the gloo tests of the bucket / step logic)
            self.optimizer = torch.optim.RMSprop(param_groups(model, weight_decay), lr=lr, alpha=alpha)
        self.group = group

    def __call__(self, batch, img_scales, inter_scales, is_flow=True):
        """batch: the reference's data_batch (img_list, cam_params_list, mean, std, gt_depth_img) on the device.
        Returns (total loss (detached), loss dict, preds)."""
        self.model.train()
        self.bucket.zero_()                                        # optimizer.zero_grad(), keeping the views
        with train_ops.direct_grads():                             # the fused nodes add into the bucket themselves
            preds = self.model(batch, img_scales, inter_scales, isFlow=is_flow, isTest=False)
            losses = self.loss_fn(preds, batch, is_flow)
            total = sum(losses.values())
            total.backward()
        join_fork_streams()                                        # the flow tower's backward ran beside the coarse stage's
        self.bucket.allreduce_sum(self.group)                      # SUM: the loss sums over the batch (networks.py:176-179)
        self.optimizer.step()
        return total.detach(), losses, preds


class GraphedTrainStep(object):
    """The same step with  zero_grad + forward + loss + backward  captured ONCE in a hipGraph and replayed per scene.

    Eager, config 4 spends ~100 of its 117 ms per step in Python / ATen launch overhead (autograd over ~1 500 small
    kernels); the GPU work itself is a fraction of that.  What makes the pass capturable: every host-derived constant
    lives in a TrainPlan (model.py; one pinned block, one H2D per step, OUTSIDE the graph), the images / ground
    truth / camera block are static device buffers the step's batch is copied into, the gradients already live in
    the flat bucket (GradBucket: ``.grad`` views, zeroed by one memset inside the graph), and the packed weights of
    the fused EdgeConv node are re-packed inside the graph (pointflow.no_pack_cache) because the parameters change
    between replays.  The gradient all-reduce (one collective) and the RMSprop step (a handful of foreach kernels)
    run after the replay, eagerly -- RCCL calls are kept out of the graph on purpose.

    Warm-up forwards/backwards run on a side stream before the capture (library autotuning, allocator); the
    BatchNorm buffers are restored afterwards, the parameters are not touched (no optimizer step during warm-up)."""

    def __init__(self, trainer, batch, img_scales, inter_scales, is_flow=True, warmup=3):
        self.t = trainer
        self.is_flow = bool(is_flow)
        model = trainer.model
        model.train()
        self.img = batch["img_list"].detach().clone()
        self.gt = batch["gt_depth_img"].detach().clone()
        self.cams = batch["cam_params_list"].detach().clone()
        self.plan = model.make_train_plan(batch, img_scales, inter_scales, isTest=False)
        buffers = [(b, b.detach().clone()) for b in model.buffers()]
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=self.img.device)
        side.wait_stream(main)
        with pointflow.no_pack_cache():
            with torch.cuda.stream(side):
                for _ in range(int(warmup)):
                    self._forward_backward()
            main.wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():
                for b, saved in buffers:
                    b.copy_(saved)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.total, self.losses, self.preds = self._forward_backward()

    def _forward_backward(self):
        self.t.bucket.zero_()
        with train_ops.direct_grads():
            preds = self.t.model.run_autograd(self.plan, self.img, self.is_flow)
            labels = {"gt_depth_img": self.gt, "cam_params_list": self.cams}
            losses = self.t.loss_fn(preds, labels, self.is_flow)
            total = sum(losses.values())
            total.backward()
        join_fork_streams()
        return total.detach(), {k: v.detach() for k, v in losses.items()}, {k: v.detach() for k, v in preds.items()}

    def __call__(self, batch):
        """One step on ``batch`` (same shapes as the capture batch).  Returns (total loss, loss dict, preds) -- static
        tensors that the next call overwrites."""
        self.plan.update_(batch)                                    # host camera algebra + one H2D
        self.img.copy_(batch["img_list"], non_blocking=True)
        self.gt.copy_(batch["gt_depth_img"], non_blocking=True)
        self.cams.copy_(batch["cam_params_list"], non_blocking=True)
        self.graph.replay()
        self.t.bucket.allreduce_sum(self.t.group)
        self.t.optimizer.step()
        return self.total, self.losses, self.preds
