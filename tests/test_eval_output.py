"""The step after the path: evaluation files and the fusion pre-step, byte for byte
against files written by the reference's own writers (tests/golden/make_eval_golden.py runs reference
utils/eval_file_logger.py, utils/io.py and tools/depthfusion.py::probability_filter)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from pointmvsnet_amd.utils import eval_file_logger as E
from pointmvsnet_amd.utils import io as IO


def _golden():
    g = np.load(os.path.join(GOLDEN_DIR, "eval_output.npz"))
    preds = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("pred:")}
    # dict order matters to the reference (it iterates preds.keys()): coarse, then flow1_prob, flow1, flow2_prob, flow2
    order = ["coarse_depth_map", "coarse_prob_map", "flow1_prob", "flow1", "flow2_prob", "flow2"]
    preds = {k: preds[k] for k in order}
    files = {k[5:]: g[k].tobytes() for k in g.files if k.startswith("file:")}
    batch = {"cam_params_list": torch.from_numpy(g["cam_params_list"]), "img_list": torch.zeros(1, 3, 3, 128, 192)}
    return preds, files, batch


def _check_logger_files(scene, files):
    names = sorted(k[7:] for k in files if k.startswith("logger/"))
    assert sorted(os.listdir(scene)) == names
    for n in names:
        assert open(os.path.join(scene, n), "rb").read() == files["logger/" + n], n


def test_host_logger_and_filter_write_the_reference_bytes(tmp_path):
    preds, files, batch = _golden()
    ref_path = str(tmp_path / "Eval" / "Rectified" / "scan9" / "rect_004_3_r5000.png")
    E.eval_file_logger(batch, preds, ref_path, "out")
    scene = str(tmp_path / "Eval" / "out" / "scan9")
    _check_logger_files(scene, files)
    for n in os.listdir(scene):                                    # the filter reads view-indexed names
        os.link(os.path.join(scene, n), os.path.join(scene, "00000000" + n[8:]) if n[0] == "0" else
                os.path.join(scene, "x" + n))
    E.probability_filter(scene, 0.2, 0.1, "flow2", 1, "NEAREST")
    assert open(os.path.join(scene, "00000000_flow2_prob_filtered.pfm"), "rb").read() == files["filter/flow2_nearest"]
    E.probability_filter(scene, 0.2, 0.1, "flow1", 1, "NEAREST")
    assert open(os.path.join(scene, "00000000_flow1_prob_filtered.pfm"), "rb").read() == files["filter/flow1_same"]
    with pytest.raises(NotImplementedError):
        E.probability_filter(scene, 0.2, 0.1, "flow1", 1, "LANCZOS4")


def test_pfm_and_cam_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    img = torch.rand(7, 5, generator=g).numpy()
    IO.write_pfm(str(tmp_path / "a.pfm"), img)
    back, scale = IO.load_pfm(str(tmp_path / "a.pfm"))
    assert np.array_equal(back, img) and scale == 1.0
    # writable like the reference's np.fromfile result: depthfusion.py masks the loaded depth map in place
    back[back < 2.0] = 0.0
    assert float(back.max()) == 0.0
    IO.write_pfm_body(str(tmp_path / "b.pfm"), np.ascontiguousarray(img[::-1]).tobytes(), 5, 7)
    assert open(str(tmp_path / "a.pfm"), "rb").read() == open(str(tmp_path / "b.pfm"), "rb").read()
    with pytest.raises(Exception):
        IO.write_pfm(str(tmp_path / "c.pfm"), img.astype(np.float64))
    cam = np.arange(32, dtype=np.float32).reshape(2, 4, 4) * 1.25
    IO.write_cam_dtu(str(tmp_path / "cam.txt"), cam)
    got = IO.load_cam_dtu(open(str(tmp_path / "cam.txt")))
    assert np.array_equal(got[0], cam[0]) and np.array_equal(got[1, :3, :3], cam[1, :3, :3])
    assert np.array_equal(got[1, 3], cam[1, 3])


@pytest.mark.gpu
def test_device_packing_and_async_writer_write_the_reference_bytes(dev, tmp_path):
    """The device path: packed staging buffer, asynchronous D2H, background writer -- same bytes, plus the
    probability-filtered depth maps straight from the device."""
    preds, files, batch = _golden()
    preds_dev = {k: v.to(dev) for k, v in preds.items()}
    batch_dev = {"cam_params_list": batch["cam_params_list"].to(dev), "img_list": batch["img_list"],
                 "cam_params_list_host": batch["cam_params_list"]}
    ref_path = str(tmp_path / "Eval" / "Rectified" / "scan9" / "rect_004_3_r5000.png")
    w = E.AsyncEvalWriter(filter_thresholds=(0.2, 0.1))
    for _ in range(3):                                             # several maps in flight
        w.submit(batch_dev, preds_dev, ref_path, "out")
    w.close()
    scene = str(tmp_path / "Eval" / "out" / "scan9")
    assert open(os.path.join(scene, "00000003_flow2_prob_filtered.pfm"), "rb").read() == files["filter/flow2_nearest"]
    assert open(os.path.join(scene, "00000003_flow1_prob_filtered.pfm"), "rb").read() == files["filter/flow1_same"]
    os.remove(os.path.join(scene, "00000003_flow2_prob_filtered.pfm"))
    os.remove(os.path.join(scene, "00000003_flow1_prob_filtered.pfm"))
    _check_logger_files(scene, files)
    # the synchronous entry point takes the device path too
    E.eval_file_logger(batch_dev, preds_dev, str(tmp_path / "E2" / "Rectified" / "scan9" / "rect_004_3_r5000.png"), "o")
    _check_logger_files(str(tmp_path / "E2" / "o" / "scan9"), files)


@pytest.mark.gpu
def test_async_writer_orders_the_producer_behind_its_packs(dev, tmp_path):
    """GraphedForward hands out STATIC prediction buffers that the next replay overwrites: ``submit`` must make the
    producer stream wait for the pack kernels that still read them.  The writer's side stream is stalled with a long
    sleep kernel, the maps are submitted and then immediately overwritten on the producer stream -- the files must
    still hold the submitted values."""
    preds, files, batch = _golden()
    static = {k: v.to(dev).clone() for k, v in preds.items()}
    batch_dev = {"cam_params_list": batch["cam_params_list"].to(dev), "img_list": batch["img_list"],
                 "cam_params_list_host": batch["cam_params_list"]}
    ref_path = str(tmp_path / "Eval" / "Rectified" / "scan9" / "rect_004_3_r5000.png")
    w = E.AsyncEvalWriter(filter_thresholds=(0.2, 0.1))
    w.submit(batch_dev, static, ref_path, "warm")                 # creates the side stream
    with torch.cuda.stream(w._stream):
        torch.cuda._sleep(400 * 1000 * 1000)                      # ~0.2 s: the packs of the next submit queue behind it
    w.submit(batch_dev, static, ref_path, "out")
    for v in static.values():                                      # "the next replay": same stream as the producer
        v.fill_(123.0)
    w.close()
    scene = str(tmp_path / "Eval" / "out" / "scan9")
    os.remove(os.path.join(scene, "00000003_flow2_prob_filtered.pfm"))
    os.remove(os.path.join(scene, "00000003_flow1_prob_filtered.pfm"))
    _check_logger_files(scene, files)
