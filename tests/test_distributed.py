"""CPU, world_size 2 over gloo: scene sharding and the single flat-bucket SUM all-reduce used by the
data-parallel training configuration."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointmvsnet_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                         # replicas start different on purpose
    net = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 1, bias=False), torch.nn.BatchNorm1d(8), torch.nn.Conv1d(8, 1, 1))
    D.broadcast_parameters(net, src=0)
    x = torch.full((1, 4, 6), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    n = D.allreduce_gradients_sum(net)
    gathered = [torch.zeros_like(torch.cat([g.reshape(-1) for g in local])) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([g.reshape(-1) for g in local]))
    want = sum(gathered)                                   # SUM, not mean
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    out[rank] = (n, bool(torch.allclose(got, want, rtol=1e-6, atol=1e-7)), D.shard_scenes(7, rank, world),
                 float(list(net.parameters())[0].sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_sum_and_scene_sharding_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0][1] and out[1][1]
    assert out[0][0] == out[1][0] == 4 * 8 + 8 + 8 + 8 + 1
    assert sorted(out[0][2] + out[1][2]) == list(range(7)) and out[0][2] == [0, 2, 4, 6]
    assert out[0][3] == out[1][3]                          # broadcast made the replicas identical


def _bucket_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env(backend="gloo")
    from pointmvsnet_amd import synthetic
    from pointmvsnet_amd.model import PointMVSNet
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=rank)                # replicas start different on purpose
    D.broadcast_parameters(net, src=0)
    bucket = D.GradBucket(net)
    g = torch.Generator().manual_seed(50 + rank)
    local = []
    for p in net.parameters():                            # what backward would do: accumulate into p.grad in place
        val = torch.randn(p.shape, generator=g)
        p.grad.add_(val)
        local.append(val.reshape(-1))
    local = torch.cat(local)
    n = bucket.allreduce_sum()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered)
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    first = float(next(net.parameters()).sum())
    bucket.zero_()
    zeroed = all(float(p.grad.abs().sum()) == 0.0 for p in net.parameters()) and bucket.attached()
    out[rank] = (n, bool(torch.equal(got, want)), bool(torch.equal(bucket.flat, torch.zeros_like(bucket.flat))), zeroed,
                 first)
    dist.barrier()
    dist.destroy_process_group()


def test_pointmvsnet_gradient_bucket_allreduce_world2():
    """The REAL bucket of BASELINE config 4: all 698 936 PointMVSNet parameters as views into one flat float32
    buffer, one in-place SUM all-reduce (gloo here, RCCL on the node)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        n, summed, flat_zero, zeroed, _ = out[r]
        assert n == 698936 and summed and flat_zero and zeroed
    assert out[0][4] == out[1][4]


def test_single_process_is_a_noop():
    net = torch.nn.Linear(3, 2)
    net(torch.ones(1, 3)).sum().backward()
    before = net.weight.grad.clone()
    assert D.allreduce_gradients_sum(net) == 8
    assert torch.equal(net.weight.grad, before)
    assert D.shard_scenes(5, 0, 1) == [0, 1, 2, 3, 4]


def _trainstep_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env(backend="gloo")
    from pointmvsnet_amd import synthetic
    from pointmvsnet_amd.model import PointMVSNet
    from pointmvsnet_amd.train_step import TrainStep
    net = PointMVSNet()
    synthetic.seed_weights(net, seed=rank)                # replicas start different on purpose
    mine_before = float(sum(p.detach().double().sum() for p in net.parameters()))
    step = TrainStep(net, check_every=1)                  # broadcasts rank 0's parameters and buffers
    start = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    buffers = torch.cat([b.detach().reshape(-1).double() for b in net.buffers()])
    sums = []
    g = torch.Generator().manual_seed(70 + rank)
    for _ in range(2):                                    # what a step does after its backward pass, twice
        step.bucket.zero_()
        for p in net.parameters():
            p.grad.add_(torch.randn(p.shape, generator=g) * 1e-2)      # per-rank gradients, summed by finish()
        step.finish()                                     # all-reduce(SUM) + RMSprop + the replica checksum
        sums.append(step.check_replicas())
    moved = not torch.equal(torch.cat([p.detach().reshape(-1) for p in net.parameters()]), start)
    raised = False
    if rank == 1:                                         # a replica that drifted: every rank must notice
        with torch.no_grad():
            next(net.parameters()).add_(1e-3)
    try:
        step.check_replicas()
    except RuntimeError as exc:
        raised = "diverged" in str(exc)
    out[rank] = (mine_before, float(start.double().sum()), float(buffers.sum()), sums, moved, raised, step.steps_done)
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_broadcasts_rank0_and_checks_replicas_world2():
    """TrainStep in a world of two (gloo): the replicas are seeded differently, TrainStep's constructor makes them rank
    0's model (parameters and buffers; reference train.py:177: nn.DataParallel broadcasts replica 0 every iteration),
    two finish() calls (SUM all-reduce of per-rank gradients + RMSprop) keep them equal -- the checksum all-reduce
    agrees -- and a deliberately perturbed replica makes check_replicas() raise on EVERY rank."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainstep_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0][0] != out[1][0]                          # they really started different
    assert abs(out[0][1] - out[0][0]) < 1e-9 * max(1.0, abs(out[0][0]))     # ... and both became rank 0's model
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert out[0][3] == out[1][3] and len(out[0][3]) == 2 and out[0][3][0] != out[0][3][1]
    assert out[0][4] and out[1][4]
    assert out[0][5] and out[1][5]
    assert out[0][6] == out[1][6] == 2


def test_replica_check_is_a_noop_for_one_process():
    t = torch.arange(5, dtype=torch.float32)
    assert D.assert_replicas_equal(t) == (10.0, 30.0) and D.world_size() == 1
