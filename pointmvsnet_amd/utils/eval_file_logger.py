"""Evaluation output of one depth map + the fusion pre-step.

``eval_file_logger(data_batch, preds, ref_img_path, folder)`` keeps the reference's signature and writes the
same files with the same bytes (reference utils/eval_file_logger.py:12-79): ``%08d_init.pfm``,
``%08d_init_prob.pfm``, ``cam_%08d_init.txt`` and per PointFlow iteration ``%08d_flowK.pfm``,
``%08d_flowK_prob.pfm``, ``cam_%08d_flowK.txt``, ``%08d_flowKpts.xyz`` (the .jpg copy of the reference image
needs OpenCV and is written only when cv2 is importable).

What differs is where the work happens.  The reference calls ``.cpu().numpy()`` per map -- a device
synchronisation each -- and derives the flow confidence maps with NumPy fancy indexing on the host.  Here the
device packs every map of a depth map, flow confidences and (optionally) probability-filtered depths included,
into ONE staging buffer in PFM row order (csrc/eval_out.hip), one asynchronous D2H copy on a side stream moves it
into pinned memory, and a writer thread turns it into files while the GPU is already on the next scene
(``AsyncEvalWriter``).  ``probability_filter`` is the file-based pre-step of reference tools/depthfusion.py:153-170.
"""
import os
import os.path as osp
import queue
import threading

import numpy as np
import torch

from .. import _lib
from .io import cam_dtu_text, load_pfm, mkdir, write_pfm, write_pfm_body

_OFFSETS = np.array([-2.0, -1.0, 0.0, 1.0, 2.0]).reshape(1, 1, -1)


def flow_confidence_np(prob_hw5):
    """Host statement of eval_file_logger.py:48-62 (used for CPU tensors and by the tests)."""
    idx = np.sum(prob_hw5 * _OFFSETS, axis=-1) + 2.0
    lo = np.floor(idx).astype(int)
    hi = np.clip(lo + 1, 0, prob_hw5.shape[-1] - 1)
    return np.take_along_axis(prob_hw5, lo[..., None], axis=-1)[..., 0] + \
        np.take_along_axis(prob_hw5, hi[..., None], axis=-1)[..., 0]


def depth_to_points(depth_map, cam_intrinsic, cam_extrinsic):
    """(h*w, 3) world points of a depth map (eval_file_logger.py:82-94): pixel centres through K^-1 and [R|t]^-1,
    with NumPy's own dtype promotion (float32 inverses applied to float64 grids)."""
    h, w = depth_map.shape
    xs, ys = np.meshgrid(np.linspace(0.5, w - 0.5, w), np.linspace(0.5, h - 0.5, h))
    grid = np.concatenate([xs.reshape(1, -1), ys.reshape(1, -1), np.ones((1, h * w))], axis=0)
    cam_points = np.matmul(np.linalg.inv(cam_intrinsic), grid) * np.reshape(depth_map, (1, -1))
    R, t = cam_extrinsic[:3, :3], cam_extrinsic[:3, 3:4]
    return np.matmul(np.linalg.inv(R), cam_points - t).transpose()


def _scene_paths(ref_img_path, folder):
    parts = ref_img_path.split("/")
    scene_folder = osp.join("/".join(parts[:-3]), folder, parts[-2])
    return scene_folder, int(parts[-1][5:8]) - 1


class _Job(object):
    __slots__ = ("event", "host", "layout", "cams", "ref_h", "scene_folder", "index", "ref_image", "points")


class AsyncEvalWriter(object):
    """Writes the evaluation files of depth maps in the background.

    ``submit`` enqueues the device-side packing on a side stream that waits for the producer stream, starts one
    asynchronous D2H copy into a pinned buffer and returns; a worker thread waits for the copy's event and writes
    the files.  ``filter_thresholds=(init_prob_threshold, flow_prob_threshold)`` additionally emits
    ``%08d_flowK_prob_filtered.pfm`` (the output of depthfusion.py's probability_filter with nearest resizing of
    the initial confidence) straight from the device.  ``close()`` drains the queue and re-raises a worker error."""

    def __init__(self, filter_thresholds=None, write_points=True, max_pending=8):
        self.filter_thresholds = filter_thresholds
        self.write_points = write_points
        self._queue = queue.Queue(maxsize=max_pending)
        self._error = None
        self._stream = None
        self._thread = threading.Thread(target=self._worker, daemon=True)
        self._thread.start()

    # ---- producer side ----------------------------------------------------------------------------------
    def submit(self, data_batch, preds, ref_img_path, folder):
        if self._error is not None:
            raise self._error
        coarse = preds["coarse_depth_map"]
        dev = coarse.device
        scene_folder, index = _scene_paths(ref_img_path, folder)
        flows = [k for k in preds.keys() if "flow" in k and "prob" not in k]
        # layout of the staging buffer: (name, height, width, offset)
        layout, off = [], 0

        def slot(name, t):
            nonlocal off
            h, w = int(t.shape[-2]), int(t.shape[-1])
            layout.append((name, h, w, off))
            off += h * w
            return off - h * w

        items = [("init", coarse[0, 0], None), ("init_prob", preds["coarse_prob_map"][0, 0], None)]
        for k in flows:
            items.append((k, preds[k][0, 0], None))
            if k + "_prob" in preds:
                items.append((k + "_prob", None, preds[k + "_prob"][0]))
        job = _Job()
        if dev.type != "cuda":
            raise RuntimeError("AsyncEvalWriter packs on the device; for CPU tensors call eval_file_logger_host")
        if self._stream is None or self._stream.device != dev:
            self._stream = torch.cuda.Stream(device=dev)
        side = self._stream
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.device(dev), torch.cuda.stream(side):
            sizes = sum(int(t.shape[-2]) * int(t.shape[-1]) for _, t, p in items for t in [t if t is not None else p])
            extra = sum(int(preds[k].shape[-2]) * int(preds[k].shape[-1]) for k in flows
                        if self.filter_thresholds is not None and k + "_prob" in preds)
            staging = torch.empty(sizes + extra, dtype=torch.float32, device=dev)
            conf_dev = {}
            for name, plane, prob in items:
                src = (plane if plane is not None else prob).contiguous().float()
                o = slot(name, src)
                h, w = layout[-1][1], layout[-1][2]
                dst = staging[o:o + h * w]
                if plane is not None:
                    _lib.call("pf_eval_pack_map_f32", _lib.ptr(src), _lib.ptr(dst), h, w, 1, _lib.stream())
                else:
                    _lib.call("pf_eval_flow_prob_f32", _lib.ptr(src), _lib.ptr(dst), h, w, 1, _lib.stream())
                    if self.filter_thresholds is not None:      # un-flipped copy for the filter below
                        conf = torch.empty((h, w), dtype=torch.float32, device=dev)
                        _lib.call("pf_eval_flow_prob_f32", _lib.ptr(src), _lib.ptr(conf), h, w, 0, _lib.stream())
                        conf_dev[name[:-5]] = conf
                for t in (src,):
                    t.record_stream(side)
            if self.filter_thresholds is not None:
                init_thr, flow_thr = self.filter_thresholds
                init_conf = preds["coarse_prob_map"][0, 0].contiguous().float()
                for k in flows:
                    if k not in conf_dev:
                        continue
                    depth = preds[k][0, 0].contiguous().float()
                    o = slot(k + "_prob_filtered", depth)
                    h, w = layout[-1][1], layout[-1][2]
                    _lib.call("pf_eval_prob_filter_f32", _lib.ptr(depth), _lib.ptr(conf_dev[k]), _lib.ptr(init_conf), h, w,
                              int(init_conf.shape[0]), int(init_conf.shape[1]), float(flow_thr), float(init_thr),
                              _lib.ptr(staging[o:o + h * w]), 1, _lib.stream())
            # every read of ``preds`` is enqueued: the producer stream must not overwrite them before these packs ran
            # (GraphedForward / GraphedTrainStep hand out STATIC buffers that the next replay writes; record_stream only
            # talks to the allocator).  The packs take microseconds; the D2H copy below reads the staging buffer only.
            packed = torch.cuda.Event()
            packed.record(side)
            host = torch.empty(staging.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(staging, non_blocking=True)
            staging.record_stream(side)
            job.event = torch.cuda.Event()
            job.event.record(side)
        torch.cuda.current_stream(dev).wait_event(packed)
        cams = data_batch.get("cam_params_list_host")
        if cams is None:
            cams = data_batch["cam_params_list"].detach().cpu()
        job.cams = cams.numpy()[0, 0].copy()
        ref = data_batch.get("ref_img")
        job.ref_image = None if ref is None else ref[0].detach().cpu().numpy()
        job.ref_h = int(job.ref_image.shape[0]) if job.ref_image is not None else int(data_batch["img_list"].shape[3])
        job.host, job.layout, job.scene_folder, job.index = host, layout, scene_folder, index
        job.points = self.write_points
        self._queue.put(job)

    def close(self):
        self._queue.put(None)
        self._thread.join()
        if self._error is not None:
            raise self._error

    # ---- consumer side ----------------------------------------------------------------------------------
    def _worker(self):
        while True:
            job = self._queue.get()
            if job is None:
                return
            try:
                job.event.synchronize()
                _write_files(job)
            except Exception as exc:          # surfaced by the next submit() / close()
                self._error = exc


def _write_files(job):
    mkdir(job.scene_folder)
    buf = job.host.numpy()
    prefix = osp.join(job.scene_folder, "%08d" % job.index)
    if job.ref_image is not None:
        try:
            import cv2
            cv2.imwrite(prefix + ".jpg", job.ref_image)
        except ImportError:
            pass
    for name, h, w, off in job.layout:
        body = buf[off:off + h * w]
        write_pfm_body("%s_%s.pfm" % (prefix, name), body.tobytes(), w, h)
        if name == "init" or (name.startswith("flow") and "prob" not in name):
            cam = job.cams.copy()
            cam[1, :2, :3] *= (float(h) / float(job.ref_h))
            with open(osp.join(job.scene_folder, "cam_%08d_%s.txt" % (job.index, name)), "w") as f:
                f.write(cam_dtu_text(cam))
            if name != "init" and job.points:
                depth = body.reshape(h, w)[::-1]
                pts = depth_to_points(depth, cam[1][:3, :3], cam[0])
                np.savetxt("%s_%spts.xyz" % (prefix, name), pts, delimiter=" ", fmt="%.4f")


def eval_file_logger(data_batch, preds, ref_img_path, folder):
    """The reference's synchronous entry point (same signature, same files)."""
    if preds["coarse_depth_map"].is_cuda:
        w = AsyncEvalWriter()
        w.submit(data_batch, preds, ref_img_path, folder)
        w.close()
        return
    eval_file_logger_host(data_batch, preds, ref_img_path, folder)


def eval_file_logger_host(data_batch, preds, ref_img_path, folder):
    """Pure NumPy version for CPU tensors (also the statement the device path is tested against)."""
    scene_folder, index = _scene_paths(ref_img_path, folder)
    mkdir(scene_folder)
    cams = data_batch["cam_params_list"].cpu().numpy()[0, 0]
    ref = data_batch.get("ref_img")
    ref_h = int(ref.shape[1]) if ref is not None else int(data_batch["img_list"].shape[3])
    prefix = osp.join(scene_folder, "%08d" % index)

    def cam_file(name, h):
        cam = cams.copy()
        cam[1, :2, :3] *= (float(h) / float(ref_h))
        with open(osp.join(scene_folder, "cam_%08d_%s.txt" % (index, name)), "w") as f:
            f.write(cam_dtu_text(cam))
        return cam

    init = preds["coarse_depth_map"].cpu().numpy()[0, 0]
    write_pfm(prefix + "_init.pfm", init)
    write_pfm(prefix + "_init_prob.pfm", preds["coarse_prob_map"].cpu().numpy()[0, 0])
    cam_file("init", init.shape[0])
    for k in preds.keys():
        if "flow" not in k:
            continue
        if "prob" in k:
            write_pfm("%s_%s.pfm" % (prefix, k), flow_confidence_np(preds[k][0].cpu().permute(1, 2, 0).numpy()))
        else:
            depth = preds[k][0, 0].cpu().numpy()
            write_pfm("%s_%s.pfm" % (prefix, k), depth)
            cam = cam_file(k, depth.shape[0])
            np.savetxt("%s_%spts.xyz" % (prefix, k), depth_to_points(depth, cam[1][:3, :3], cam[0]), delimiter=" ",
                       fmt="%.4f")


def _resize_nearest(img, h, w):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST): src = min(floor(dst * src/dst), src - 1)."""
    ys = np.minimum(np.floor(np.arange(h) * (img.shape[0] / float(h))).astype(int), img.shape[0] - 1)
    xs = np.minimum(np.floor(np.arange(w) * (img.shape[1] / float(w))).astype(int), img.shape[1] - 1)
    return img[ys][:, xs]


def probability_filter(scene_folder, init_prob_threshold, flow_prob_threshold, name, view_num, mode="NEAREST"):
    """The fusion pre-step of reference tools/depthfusion.py:153-170 on the files eval_file_logger wrote: depth := 0
    where the flow or the (resized) initial confidence is below its threshold -> ``%08d_<name>_prob_filtered.pfm``.
    Only nearest resizing is built (the other cv2 interpolation modes are OpenCV's own fixed-point kernels)."""
    if mode not in ("NEAREST", 0):
        raise NotImplementedError("probability_filter: only cv2.INTER_NEAREST resizing is implemented")
    for v in range(view_num):
        depth_map = load_pfm(osp.join(scene_folder, "{:08d}_{}.pfm".format(v, name)))[0].copy()
        prob_map = load_pfm(osp.join(scene_folder, "{:08d}_{}_prob.pfm".format(v, name)))[0]
        init_prob_map = load_pfm(osp.join(scene_folder, "{:08d}_init_prob.pfm".format(v)))[0]
        if prob_map.shape != depth_map.shape:
            prob_map = _resize_nearest(prob_map, *depth_map.shape)
        if init_prob_map.shape != depth_map.shape:
            init_prob_map = _resize_nearest(init_prob_map, *depth_map.shape)
        depth_map[prob_map < flow_prob_threshold] = 0
        depth_map[init_prob_map < init_prob_threshold] = 0
        write_pfm(osp.join(scene_folder, "{:08d}_{}_prob_filtered.pfm".format(v, name)), depth_map)
