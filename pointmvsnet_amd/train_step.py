"""One data-parallel training step of BASELINE config 4 (reference train.py:46-112, solver.py:17-52).

The reference's loop body is  preds = model(batch); optimizer.zero_grad(); loss = sum(loss_fn(...));
loss.backward(); optimizer.step()  under nn.DataParallel with RMSprop(lr 1e-3, alpha 0.9), weight decay off the
``.bn.`` parameters.  Here one process drives one GPU with one scene; replicas exchange gradients with ONE
in-place SUM all-reduce of the flat bucket every parameter's ``.grad`` is a view of (distributed.GradBucket).
Every convolution / BatchNorm / warp / head of the step, forward and backward, runs on this package's own kernels
behind the autograd nodes of train_ops.py (no library convolution, BatchNorm or GEMM kernel, no float atomic: the
gradient is bit-reproducible); the nodes add their parameter gradients straight into the bucket.

Replica consistency (reference train.py:177: nn.DataParallel re-broadcasts rank 0's parameters every iteration):
``TrainStep`` broadcasts parameters and buffers from rank 0 once, when it is built in a world of more than one rank --
from then on every replica applies the same all-reduced gradient with the same optimizer state, so the parameters stay
equal bit for bit -- and every ``check_every`` steps all-reduces a checksum of the flat parameter buffer (MIN and MAX
must agree) so that a replica that drifted (a rank that loaded other weights, a skipped step) fails loudly instead of
training a different model.  BatchNorm running statistics are per replica by design (DataParallel keeps replica 0's).
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


class FlatRMSprop(torch.optim.Optimizer):
    """torch.optim.RMSprop(param_groups(model, weight_decay), lr, alpha) (reference solver.py:17-52) as ONE launch per
    step (pf_rmsprop_f32): the parameters move into one flat float32 buffer -- every ``p.data`` becomes a view of it, in
    the bucket's order, so parameter i's gradient sits at the same offset of ``bucket.flat`` -- beside a flat
    square-average buffer.  PyTorch's foreach form is five multi-tensor launches over 115 tensors plus their host-side
    grouping, the only per-step work outside the captured graph.

    It IS a ``torch.optim.Optimizer``: ``param_groups`` are the reference's two groups in the reference's order (decay:
    everything but the ``.bn.`` parameters; then the ``.bn.`` ones, solver.py:33-52), so the reference's LR schedulers
    attach to it (solver.py:65-80; the kernel takes one lr / alpha / eps per step: all groups must agree, which is what
    ``build_optimizer`` and ``StepLR`` produce) and ``state_dict()`` / ``load_state_dict()`` speak torch's RMSprop
    layout -- ``state[i] = {"step", "square_avg"}`` indexed over [decay parameters..., .bn. parameters...], plus
    ``param_groups`` -- so optimizer checkpoints are interchangeable with ``torch.optim.RMSprop`` on the same groups
    (the CPU TrainStep, the reference's checkpointer)."""

    def __init__(self, bucket, named_parameters, lr=1e-3, alpha=0.9, eps=1e-8, weight_decay=0.0):
        named_parameters = list(named_parameters)
        names = {id(p): n for n, p in named_parameters}
        params = bucket.params
        # torch.optim.RMSprop on the reference's param_groups keeps EVERY parameter and both groups; this class indexes its
        # state over the bucket (the trainable parameters).  The two layouts agree only when nothing is frozen: say so here
        # instead of failing in load_state_dict on group sizes.
        held = {id(p) for p in params}
        frozen = [n for n, p in named_parameters if id(p) not in held]
        if frozen:
            raise ValueError("FlatRMSprop: every parameter must be trainable (its state layout is torch.optim.RMSprop's over "
                             "ALL parameters); not in the gradient bucket: %s" % ", ".join(frozen[:5]))
        is_bn = [".bn." in names.get(id(p), "") for p in params]
        groups = [dict(params=[p for p, bn in zip(params, is_bn) if not bn], weight_decay=float(weight_decay)),
                  dict(params=[p for p, bn in zip(params, is_bn) if bn], weight_decay=0.0)]
        defaults = dict(lr=float(lr), alpha=float(alpha), eps=float(eps), weight_decay=float(weight_decay), momentum=0.0,
                        centered=False)
        super(FlatRMSprop, self).__init__([g for g in groups if g["params"]], defaults)
        self.bucket = bucket
        self.device = bucket.flat.device
        self.flat = torch.empty_like(bucket.flat)
        self.square_avg = torch.zeros_like(bucket.flat)
        self.steps = 0
        self._span = {}                                     # id(parameter) -> (offset, numel) in the flat buffers
        offset = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[offset:offset + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[offset:offset + n].view(p.shape)          # (packed-weight caches re-pack: new storage)
                self._span[id(p)] = (offset, n)
                offset += n
        self.wd, self._wd_key = None, None
        self._refresh_weight_decay()

    def _refresh_weight_decay(self):
        """The per-element weight-decay vector of the kernel, rebuilt when a group's ``weight_decay`` changed."""
        key = tuple(float(g["weight_decay"]) for g in self.param_groups)
        if key == self._wd_key:
            return
        self._wd_key = key
        if not any(key):
            self.wd = None
            return
        self.wd = torch.zeros_like(self.flat)
        for g in self.param_groups:
            for p in g["params"]:
                o, n = self._span[id(p)]
                self.wd[o:o + n] = float(g["weight_decay"])

    def _hyper(self):
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if (g["lr"], g["alpha"], g["eps"]) != (g0["lr"], g0["alpha"], g0["eps"]):
                raise NotImplementedError("FlatRMSprop: one lr / alpha / eps for all groups (the reference's solver)")
        if any(g.get("momentum", 0.0) != 0.0 or g.get("centered", False) for g in self.param_groups):
            raise NotImplementedError("FlatRMSprop: momentum / centered RMSprop is not built (reference: plain RMSprop)")
        if any(g.get("maximize", False) for g in self.param_groups):       # (a loaded checkpoint may carry the key)
            raise NotImplementedError("FlatRMSprop: maximize=True is not built (reference: plain RMSprop)")
        return float(g0["lr"]), float(g0["alpha"]), float(g0["eps"])

    # kept as attributes for callers of the round-4 interface
    lr = property(lambda self: self._hyper()[0])
    alpha = property(lambda self: self._hyper()[1])
    eps = property(lambda self: self._hyper()[2])

    def set_lr(self, lr):
        for g in self.param_groups:
            g["lr"] = float(lr)

    def attached(self):
        base = self.flat.untyped_storage().data_ptr()
        return all(p.data.untyped_storage().data_ptr() == base for p in self.bucket.params)

    def zero_grad(self, set_to_none=False):
        """The gradients are views of the bucket and must stay attached: zero them in place, whatever is asked."""
        self.bucket.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.attached():
            raise RuntimeError("FlatRMSprop: a parameter's storage was replaced (module.to() / load on a new tensor?); "
                               "build a new TrainStep")
        lr, alpha, eps = self._hyper()
        self._refresh_weight_decay()
        from . import _lib
        with torch.cuda.device(self.device):
            _lib.call("pf_rmsprop_f32", _lib.ptr(self.flat), _lib.ptr(self.bucket.flat), _lib.ptr(self.square_avg),
                      _lib.ptr(self.wd), self.flat.numel(), lr, alpha, eps, _lib.stream(),
                      algo_bytes=20.0 * self.flat.numel())
        self.steps += 1
        # torch's optimizers update parameters through in-place ops, which the packed-weight caches watch through the
        # version counter; the flat kernel writes behind autograd's back, so tick the counters
        for p in self.bucket.params:
            torch.autograd.graph.increment_version(p)
        return loss

    def _indexed(self):
        """[(torch's parameter index, parameter)]: groups in order, parameters in group order."""
        out, i = [], 0
        for g in self.param_groups:
            for p in g["params"]:
                out.append((i, p))
                i += 1
        return out

    def state_dict(self):
        state, groups, i = {}, [], 0
        for idx, p in self._indexed():
            o, n = self._span[id(p)]
            state[idx] = {"step": torch.tensor(float(self.steps)),
                          "square_avg": self.square_avg[o:o + n].view(p.shape).clone()}
        for g in self.param_groups:
            packed = {k: v for k, v in g.items() if k != "params"}
            packed["params"] = list(range(i, i + len(g["params"])))
            i += len(g["params"])
            groups.append(packed)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts torch.optim.RMSprop's layout on the same groups (the reference's checkpoints, the CPU TrainStep)."""
        groups = sd["param_groups"]
        if [len(g["params"]) for g in groups] != [len(g["params"]) for g in self.param_groups]:
            raise ValueError("FlatRMSprop.load_state_dict: parameter groups of other sizes %r (here %r)"
                             % ([len(g["params"]) for g in groups], [len(g["params"]) for g in self.param_groups]))
        by_index = {}
        for g_sd, g in zip(groups, self.param_groups):
            for i, p in zip(g_sd["params"], g["params"]):
                by_index[i] = p
            for k, v in g_sd.items():
                if k != "params":
                    g[k] = v
        steps = []
        with torch.no_grad():
            for i, st in sd["state"].items():
                p = by_index[int(i)]
                o, n = self._span[id(p)]
                if st["square_avg"].numel() != n:
                    raise ValueError("FlatRMSprop.load_state_dict: state %s has %d elements, the parameter %d"
                                     % (i, st["square_avg"].numel(), n))
                self.square_avg[o:o + n].copy_(st["square_avg"].reshape(-1))
                if "step" in st:
                    steps.append(int(float(st["step"])))
        if steps:
            self.steps = max(steps)
        self._hyper()
        self._refresh_weight_decay()


class TrainStep(object):
    def __init__(self, model, valid_threshold=8.0, lr=1e-3, alpha=0.9, weight_decay=0.0, group=None, check_every=100):
        self.model = model
        self.loss_fn = PointMVSNetLoss(valid_threshold)            # reference config.py MODEL.VALID_THRESHOLD
        self.group = group
        self.check_every, self.steps_done = int(check_every), 0
        if distributed.world_size(group) > 1:                      # every replica starts from rank 0's model
            distributed.broadcast_parameters(model, src=0, group=group)
        self.bucket = distributed.GradBucket(model)
        if self.bucket.flat.is_cuda:
            self.optimizer = FlatRMSprop(self.bucket, list(model.named_parameters()), lr=lr, alpha=alpha,
                                         weight_decay=weight_decay)
        else:                                                      # (CPU: the gloo tests of the bucket / step logic)
            self.optimizer = torch.optim.RMSprop(param_groups(model, weight_decay), lr=lr, alpha=alpha)

    def finish(self):
        """The part of a step after the backward pass: one SUM all-reduce of the bucket, the optimizer step, and every
        ``check_every`` steps the replica checksum."""
        self.bucket.allreduce_sum(self.group)                      # SUM: the loss sums over the batch (networks.py:176-179)
        self.optimizer.step()
        self.steps_done += 1
        if (self.check_every > 0 and self.steps_done % self.check_every == 0
                and distributed.world_size(self.group) > 1):       # (one process: nothing to compare, no host sync)
            self.check_replicas()

    def check_replicas(self):
        """Raise unless every rank holds the same parameters (bit for bit: the checksum is a float64 sum of the float32
        values and of their squares, equal on ranks that applied the same updates in the same order).  A no-op for one
        process.  Returns the checksum."""
        flat = getattr(self.optimizer, "flat", None)
        if flat is None:
            flat = torch.cat([p.detach().reshape(-1) for p in self.bucket.params])
        return distributed.assert_replicas_equal(flat, self.group)

    def __call__(self, batch, img_scales, inter_scales, is_flow=True):
        """batch: the reference's data_batch (img_list, cam_params_list, mean, std, gt_depth_img) on the device.
        Returns (total loss (detached), loss dict, preds)."""
        self.model.train()
        self.bucket.zero_()                                        # optimizer.zero_grad(), keeping the views
        with train_ops.direct_grads():                             # the fused nodes add into the bucket themselves
            preds = self.model(batch, img_scales, inter_scales, isFlow=is_flow, isTest=False)
            losses = self.loss_fn(preds, batch, is_flow)
            total = _total(losses)
            total.backward()
        join_fork_streams()                                        # the flow tower's backward ran beside the coarse stage's
        self.finish()
        return total.detach(), losses, preds


def _total(losses):
    """sum(losses.values()) as the reference writes it (train.py:74), without the launch that adds the int 0 first
    (0 + x is x bit for bit)."""
    vals = list(losses.values())
    total = vals[0]
    for v in vals[1:]:
        total = total + v
    return total


class GraphedTrainStep(object):
    """The same step with  zero_grad + forward + loss + backward  captured ONCE in a hipGraph and replayed per scene.

    Eager, the step is bound by Python / launch overhead (a few hundred dependent launches of 5-200 us); replayed it is
    the GPU work alone.  What makes the pass capturable: every host-derived constant
    lives in a TrainPlan (model.py; one pinned block, one H2D per step, OUTSIDE the graph), the images / ground
    truth / camera block are static device buffers the step's batch is copied into, the gradients already live in
    the flat bucket (GradBucket: ``.grad`` views, zeroed by one memset inside the graph), and the packed weights of
    the step are refreshed inside the graph (train_packs.TrainPacks: one launch) because the parameters change
    between replays.  The gradient all-reduce (one collective) and the RMSprop step (one pf_rmsprop_f32 launch over the
    flat buffers) run after the replay, eagerly -- RCCL calls are kept out of the graph on purpose.

    Warm-up forwards/backwards run on a side stream before the capture (library autotuning, allocator); the
    BatchNorm buffers are restored afterwards, the parameters are not touched (no optimizer step during warm-up)."""

    def __init__(self, trainer, batch, img_scales, inter_scales, is_flow=True, warmup=3, keep_graph=False):
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
            # keep_graph: the hipGraph itself stays reachable (raw_cuda_graph(): bench.py counts its kernel nodes)
            self.graph = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
            from .graph import capturing
            with capturing(self.graph):
                self.total, self.losses, self.preds = self._forward_backward()

    def _forward_backward(self):
        self.t.bucket.zero_()
        with train_ops.direct_grads():
            preds = self.t.model.run_autograd(self.plan, self.img, self.is_flow)
            labels = {"gt_depth_img": self.gt, "cam_params_list": self.cams}
            losses = self.t.loss_fn(preds, labels, self.is_flow)
            total = _total(losses)
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
        self.t.finish()
        return self.total, self.losses, self.preds
