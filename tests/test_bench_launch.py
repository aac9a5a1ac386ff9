"""bench.py --gpus N is a real launcher: with N > 1 and no torchrun environment it re-launches itself under
torch.distributed.run with one process per GPU (the reference fans out inside one command through nn.DataParallel,
train.py:177 / test.py:84).  CPU: the launch path with 2 gloo ranks (--launch-check stops after the rank count).
GPU: 2 RCCL ranks end to end when the box has 2 GPUs."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_gpus_flag_spawns_that_many_ranks_world2_gloo():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(CUDA_VISIBLE_DEVICES="",
                       HIP_VISIBLE_DEVICES=""), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"


def test_eight_ranks_the_drivers_scaling_run_shape_gloo():
    """`bench.py --gpus 8` as the driver's scaling run launches it: eight ranks rendezvous on 127.0.0.1, the all-reduce of
    ones counts eight, and every rank's host thread pool is capped (8 x one-thread-per-CPU would oversubscribe the node)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--launch-check"], env=_env(CUDA_VISIBLE_DEVICES="",
                       HIP_VISIBLE_DEVICES=""), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["backend"] == "gloo"
    assert 1 <= out["host_threads_per_rank"] <= 16


def test_gpus_flag_must_agree_with_the_torchrun_environment():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], env=_env(WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_launch_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5] == BENCH and cmd[-4:] == ["--gpus", "8", "--steps", "5"]


@pytest.mark.gpu
def test_two_rccl_ranks_end_to_end_when_two_gpus_are_visible():
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box; the 2-rank RCCL run needs two (the gloo test above covers the launcher)")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--scenes-per-step", "4",
                        "--calibration-steps", "2", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["value"] > 0 and out["cpu_baseline"] is None


def test_train_block_failure_costs_the_train_entry_not_the_line():
    """The train block runs in a child process (bench.train_block_in_child): here, without a GPU, the child fails -- the
    parent gets an ``error`` entry back instead of an exception, i.e. the headline line would still be printed."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(train_steps=2, train_timeout=300.0)
    saved = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") if k in os.environ}
    try:
        out = bench.train_block_in_child(args, 0, 1)
    finally:
        os.environ.update(saved)
    if torch.cuda.is_available():
        assert out.get("value", 0) > 0 or "error" in out
    else:
        assert set(out) >= {"error"} and "left no result" in out["error"], out


def test_train_block_keeps_what_a_killed_child_had_published(monkeypatch):
    """The child publishes its result part by part (timing first, then the dispatch count, then each experiment); a child
    that is killed at --train-timeout still leaves the last complete line."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench

    def fake_run(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, kw.get("timeout"), output='junk\n{"value": 1.0}\n{"value": 2.0, "experiments": {}}\n{"val')

    monkeypatch.setattr(subprocess, "run", fake_run)
    out = bench.train_block_in_child(argparse.Namespace(train_steps=2, train_timeout=1.0, no_experiments=False), 0, 1)
    assert out["value"] == 2.0 and "killed" in out["child_note"] and out["child_wall_s"] >= 0.0


def test_bench_line_assembles_on_the_emulator():
    """bench.py end to end WITHOUT a GPU (PF_EMULATE=1: tests/hipemu behind the C ABI; eager, one lane, "tiny"): scenes,
    model, the HIP-event calibration clock per entry point and per template instantiation, the timed loop and the assembly
    of the ONE JSON line all execute -- the numbers are the emulator's speed and mean nothing, the contract fields and the
    round-5 keys must be there.  (A Python error in that assembly would cost the driver's headline line.)"""
    if torch.cuda.is_available():
        pytest.skip("the emulator is for machines without a GPU")
    p = subprocess.run([sys.executable, BENCH, "--config", "tiny", "--steps", "1", "--warmup", "1", "--scenes-per-step", "1",
                        "--calibration-steps", "1"], env=_env(PF_EMULATE="1"), capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stderr[-3000:]
    out = _last_json(p.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "kernels", "train", "cpu_baseline"):
        assert key in out, key
    assert out["value"] > 0 and out["steps"] == 1 and out["dtype"] == "f32" and out["config"]["workload"].startswith("tiny")
    roof = out["roofline"]
    assert roof["kernel"].startswith("pf_conv2d_wide") and roof["bound"] in ("mfma", "hbm") and 0.0 < roof["frac"] < 1.0
    assert roof["traffic_source"] is None or "not measured in this run" in roof["traffic_source"]
    inst = roof["instantiations"]
    assert inst and {"layer", "launches", "avg_launch_us", "frac_of_f32_mfma_peak"} <= set(inst[0])
    assert any(r["layer"] == "64->64 3x3/1" for r in inst)
    assert roof["under_load"] is None or roof["under_load"]["worst"]["frac_mfma_peak"] < roof["under_load"]["best"]["frac_mfma_peak"]


def test_experiments_past_the_childs_budget_are_skipped_by_name(monkeypatch):
    """bench.experiments_block: a part that would START after PF_EXPERIMENTS_BUDGET_S seconds of the child's life is skipped
    and named, and every part publishes an updated line."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("PF_EXPERIMENTS_BUDGET_S", "0")
    out, published = {"value": 1.0}, []
    bench.experiments_block(torch.device("cpu"), out, lambda o: published.append(json.dumps(o)))
    assert len(published) == 2 and json.loads(published[-1]) == out
    assert [s.split()[0] for s in out["experiments"]["skipped"]] == ["lazy_bn", "cpu_baseline"]
    assert "cpu_baseline" not in out
