"""ctypes binding of libpointflow_hip.so (the C ABI declared in include/pointflow_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  The
product path never routes through torch eager ops or the CPU oracle for the hot-path operators.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PF_LIB_PATH: tools only -- the ablation builds of tools/experiments/build_dbg_variants.sh; the product loads the in-tree library)
LIB_PATH = os.environ.get("PF_LIB_PATH") or os.path.join(_HERE, "libpointflow_hip.so")

_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double



class BnJob(ctypes.Structure):
    """``pf_bn_job`` of include/pointflow_hip.h (one BatchNorm finalize job)."""
    _fields_ = [("partials", _vp), ("T", ctypes.c_int32), ("pcols", ctypes.c_int32), ("col0", ctypes.c_int32),
                ("C", ctypes.c_int32), ("count", _d), ("unbias_n", _d), ("gamma", _vp), ("beta", _vp),
                ("running_mean", _vp), ("running_var", _vp), ("momentum", _f), ("eps", _f),
                ("G", ctypes.c_int32), ("groups_per_stat", ctypes.c_int32), ("scale", _vp), ("shift", _vp),
                ("ld_affine", ctypes.c_int32), ("rows4", ctypes.c_int32)]


class WgradItem(ctypes.Structure):
    """``pf_wgrad_item`` of include/pointflow_hip.h (one layer of pf_conv_wgrad_batch_f32)."""
    _fields_ = [("gr", _vp), ("x", _vp)] + [(n, _i64) for n in ("N", "Cg", "Cx", "Do", "Ho", "Wo", "Di", "Hi", "Wi")] + \
               [(n, ctypes.c_int32) for n in ("KD", "KH", "KW", "stride", "pd", "ph", "pw", "x_samples_per_stat")] + \
               [("x_scale", _vp), ("x_shift", _vp), ("workspace", _vp), ("workspace_bytes", _i64)] + \
               [(n, _i64) for n in ("rows_P", "ldg", "ldx", "x_rows_per_stat")]


# name -> argtypes; every entry must exist in include/pointflow_hip.h (tests/test_abi.py checks both ways)
PROTOTYPES = {
    "pf_version": ([], ctypes.c_char_p),
    "pf_error_string": ([_i], ctypes.c_char_p),
    "pf_device_info": ([ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.c_char_p, _i], _i),
    "pf_debug_timestamp": ([_vp, _vp], _i),
    "pf_check_status": ([ctypes.POINTER(ctypes.c_uint), _vp], _i),
    "pf_gather_knn_forward_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_gather_knn_forward_f64": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_gather_knn_backward_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp], _i),
    "pf_gather_knn_backward_f64": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp], _i),
    "pf_knn_lattice_f32": ([_vp, ctypes.POINTER(_i64), _i64, _i64, _i64, _i64, _i, _i, _vp, _vp, _vp], _i),
    "pf_fetch_forward_f32": ([_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_fetch_backward_f32": ([_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_fetch_variance_f32": ([_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i, _vp], _i),
    "pf_frustum_variance_f32": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_frustum_variance_cl_f32": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_nchw_to_nhwc_f32": ([_vp, _vp, _i64, _i64, _i64, _vp], _i),
    "pf_resize_bilinear_f32": ([_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_flow_pyramid_f32": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp,
                             ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp], _i),
    "pf_flow_features_f32": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp], _i),
    "pf_stat_blocks": ([_i, _i], _i),
    "pf_gemm_blocks": ([_i, _i], _i),
    "pf_pointwise_gemm_f32": ([_vp, _i, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, ctypes.POINTER(BnJob), _i,
                               _vp, _vp], _i),
    "pf_conv3d_blocks": ([_i64, _i64, _i64, _i64, _i64, _i], _i),
    "pf_conv3d_k3_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i, _vp, _vp, ctypes.POINTER(BnJob), _i,
                          _vp, _vp], _i),
    "pf_conv3d_pair_blocks": ([_i64, _i64, _i64, _i64, _i64], _i),
    "pf_conv3d_k3_pair_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp], _i),
    "pf_conv3d_k3_few_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_deconv3d_blocks": ([_i64, _i64, _i64], _i),
    "pf_deconv3d_k3s2_f32": ([_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, ctypes.POINTER(BnJob), _i,
                              _vp, _vp], _i),
    "pf_conv3d_bottom_supported": ([_i64, _i64, _i], _i),
    "pf_conv3d_bottom_blocks": ([_i64, _i64, _i64, _i], _i),
    "pf_conv3d_bottom_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i, _vp, _vp, ctypes.POINTER(BnJob), _i,
                              _vp, _vp], _i),
    "pf_deconv3d_bottom_supported": ([_i64, _i64], _i),
    "pf_deconv3d_bottom_blocks": ([_i64, _i64, _i64], _i),
    "pf_deconv3d_bottom_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, ctypes.POINTER(BnJob), _i,
                                _vp, _vp], _i),
    "pf_conv2d_wide_supported": ([_i64, _i64, _i, _i], _i),
    "pf_conv2d_wide_blocks": ([_i64, _i64, _i64, _i], _i),
    "pf_conv2d_wide_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i, _i, _vp, _vp, ctypes.POINTER(BnJob), _i,
                            _vp, _i, _vp], _i),
    "pf_conv2d_wide_sets_f32": ([_vp, _i, _vp, _i64, _i, _vp, _i64, _i64, _i64, _i64, _i64, _i, _i, _vp, _vp,
                                 ctypes.POINTER(BnJob), _i, _vp, _i, _vp], _i),
    "pf_norm_blocks": ([_i64], _i),
    "pf_channel_stats_f32": ([_vp, _i64, _i64, _i64, _vp, _vp], _i),
    "pf_channel_affine_f32": ([_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _vp], _i),
    "pf_channel_bn_apply_f32": ([_vp, _vp, _vp, _i, _i64, _i64, _i64, _i, _d, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp], _i),
    "pf_channel_bn_apply2_f32": ([_vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _f, _vp,
                                  _i64, _i64, _i64, _i, _d, _vp], _i),
    "pf_channel_bn_fused_f32": ([_vp, _vp, _i64, _i64, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp], _i),
    "pf_edge_stats_f32": ([_vp, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp], _i),
    "pf_edge_backward_reduce_f32": ([_vp, _i64, _i, _vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "pf_edge_backward_apply_f32": ([_vp, _i64, _i, _vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                    _vp, _vp, _vp, _vp], _i),
    "pf_edge_backward_sums_f32": ([_vp, _i64, _i, _vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp,
                                   _i64, _i, _vp], _i),
    "pf_edge_backward_finish_f32": ([_vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                     _vp, _vp, _vp, _i, _vp], _i),
    "pf_knn_inverse_workspace": ([_i, _i, _i], _i64),
    "pf_knn_inverse": ([_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp], _i),
    "pf_bn_finalize_f32": ([_vp, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _i, _vp], _i),
    "pf_bn_finalize_jobs_f32": ([ctypes.POINTER(BnJob), _i, _vp], _i),
    "pf_edge_apply_f32": ([_vp, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _i, _i, _i, _vp], _i),
    "pf_flow_head_f32": ([_vp, _i64, _vp, _vp, _i, ctypes.POINTER(BnJob), _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp,
                          _vp], _i),
    "pf_eval_pack_map_f32": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "pf_eval_flow_prob_f32": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "pf_eval_prob_filter_f32": ([_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _i, _vp], _i),
    "pf_softargmin_prob_f32": ([_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp], _i),
    "pf_bn_train_rows_f32": ([_vp, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _i, _i, _i, _vp], _i),
    "pf_bn_bwd_reduce_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _vp, _vp], _i),
    "pf_softargmin_backward_f32": ([_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp], _i),
    "pf_flow_head_train_f32": ([_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp], _i),
    "pf_flow_head_backward_workspace": ([_i64], _i64),
    "pf_flow_head_backward_f32": ([_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _vp, _i64, _vp], _i),
    "pf_rmsprop_f32": ([_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _vp], _i),
    "pf_masked_mae_f32": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "pf_masked_mae_backward_f32": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "pf_wgrad_reduce_batch_f32": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp], _i),
    "pf_conv_wgrad_batch_f32": ([ctypes.POINTER(WgradItem), _i, _vp], _i),
    "pf_bn_bwd_apply_fused_f32": ([_vp, _vp, _vp, _vp, _i, _d, _vp, _i64, _i64, _i64, _i, _i, _vp, _vp, _i, _vp], _i),
    "pf_bn_bwd_plane_supported": ([_i64, _i], _i),
    "pf_bn_bwd_plane_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i, _vp, _vp, _vp, _i, _vp], _i),
    "pf_edge_backward_coeffs_f32": ([_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "pf_bn_bwd_coeffs_f32": ([_vp, _i, _i, _i, _i, _d, _i, _i, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "pf_bn_bwd_apply_f32": ([_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _vp], _i),
    "pf_rows_bn_blocks": ([_i, _i], _i),
    "pf_rows_bn_bwd_reduce_f32": ([_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "pf_rows_bn_bwd_apply_f32": ([_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp], _i),
    "pf_rows_affine_f32": ([_vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp], _i),
    "pf_conv_wgrad_workspace": ([_i64] * 9 + [_i] * 4, _i64),
    "pf_conv_wgrad_f32": ([_vp, _vp, _vp] + [_i64] * 9 + [_i] * 7 + [_vp, _vp, _i, _vp, _i64, _i, _vp], _i),
    "pf_conv_wgrad_plan": ([_i64] * 9 + [_i] * 4 + [ctypes.POINTER(_i)], _i),
    "pf_rows_wgrad_plan": ([_i64, _i, _i, ctypes.POINTER(_i)], _i),
    "pf_rows_wgrad_workspace": ([_i64, _i, _i], _i64),
    "pf_rows_wgrad_f32": ([_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _i64, _vp, _i64, _i, _vp], _i),
    "pf_deconv2d_k5s2_supported": ([_i64, _i64], _i),
    "pf_deconv2d_k5s2_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_conv3d_k3_c1_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp], _i),
    "pf_sort_pairs_workspace": ([_i64, _i64], _i64),
    "pf_sort_pairs_by_key": ([_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp], _i),
    "pf_warp_taps_flow_f32": ([_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp], _i),
    "pf_warp_taps_frustum_f32": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "pf_variance_grad_f32": ([_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _i64, _i, _vp, _vp], _i),
    "pf_warp_gather_f32": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "pf_flow_depth_grad_f32": ([_vp, _i64, _i, _vp, _i, _i, _vp, _vp], _i),
    "pf_resize_bilinear_backward_f32": ([_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "pf_pack_desc_bytes": ([], _i),
    "pf_pack_gather_f32": ([_vp, _i, ctypes.c_longlong, _vp], _i),
}

_lib = None


def load():
    """Load (once) and return the ctypes library; raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pointmvsnet_amd: %s is missing. Build it with `python -m pointmvsnet_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU or eager fallback for the hot path." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise RuntimeError("pointmvsnet_amd: cannot load %s: %s" % (LIB_PATH, exc))
    for name, (argtypes, restype) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here == ABI mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().pf_error_string(int(code)).decode()
        raise RuntimeError("pointflow_hip %s failed (%d): %s" % (what, code, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    """The current HIP stream of the current device as a void pointer.  (torch.cuda.current_stream() builds a Stream
    object through three layers of device-index helpers: 9 us per call, measured as 17 % of the drop-in route's host time
    in round 6 -- profiles/r06b_route_profile.md; the raw handle is one C call.)"""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class _NoGuard(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """``with on_device(t.device):`` = torch.cuda.device(dev) when ``dev`` is not the current device, else nothing
    (the guard's enter / exit is ~10 us of Python per operator call on the eager route)."""
    if dev.index is None or dev.index == torch._C._cuda_getDevice():
        return _NO_GUARD
    return torch.cuda.device(dev)


def require_gpu(*tensors):
    """The reference raises for non-CUDA tensors (gather_knn_kernel.cu:10,33-34); so do we."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pointmvsnet_amd: expected a GPU (HIP) tensor; this operator has no CPU path")


def status():
    """Synchronise the current stream, return and clear the sticky device status bits."""
    val = ctypes.c_uint(0)
    check(load().pf_check_status(ctypes.byref(val), stream()), "pf_check_status")
    return int(val.value)


# ---------------------------------------------------------------------------------------------
# optional per-kernel timing with HIP events (bench.py roofline leg)
# ---------------------------------------------------------------------------------------------
class KernelTimer(object):
    """Records a HIP event pair around selected C-ABI calls, on the stream the kernels are launched on
    (torch's current stream), together with the call's algorithmic byte count.  ``only`` restricts
    the instrumentation to one entry point so that the timed region of bench.py carries two event
    records per launch of the dominant kernel and nothing else."""

    def __init__(self, only=None):
        self.only = only
        self.records = []          # (name, start_event, end_event, algo_bytes, flops, tag)

    def wants(self, name):
        return self.only is None or name == self.only

    @staticmethod
    def _pair_overhead_ms():
        """Elapsed time of an EMPTY event pair on this stream (the floor every record below contains)."""
        vals = []
        for _ in range(21):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            torch.cuda.synchronize()
            vals.append(e0.elapsed_time(e1))
        return sorted(vals)[len(vals) // 2]

    def summary(self):
        torch.cuda.synchronize()
        floor = self._pair_overhead_ms()
        out = {}
        self.by_tag = {}           # (entry point, tag) -> the same sums: one row per template instantiation behind an entry
        for name, e0, e1, nbytes, flops, tag in self.records:
            ms = e0.elapsed_time(e1)                       # raw pair time; the floor is REPORTED, not subtracted
            for table, key in ((out, name), (self.by_tag, (name, tag))):
                if table is self.by_tag and tag is None:
                    continue
                s = table.setdefault(key, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0, "event_floor_ms": floor})
                s["launches"] += 1
                s["ms"] += ms
                s["bytes"] += float(nbytes or 0)
                s["flops"] += float(flops or 0)
        return out


_timer = None


def set_timer(timer):
    global _timer
    _timer = timer


_PROBE = set(filter(None, os.environ.get("PF_PROBE_DOUBLE", "").split(",")))   # sensitivity probe (tools only): run twice


def call(name, *args, **kw):
    """Invoke C-ABI entry point ``name`` and raise on a non-zero return.  ``algo_bytes`` (keyword) is the
    algorithmic HBM byte count of this launch, used only by KernelTimer."""
    algo_bytes = kw.pop("algo_bytes", None)
    flops = kw.pop("flops", None)
    tag = kw.pop("tag", None)          # which template instantiation this call selects (KernelTimer.by_tag), e.g. "64->64 3x3/1"
    fn = getattr(load(), name)
    t = _timer
    if t is not None and t.wants(name):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        code = fn(*args)
        e1.record()
        t.records.append((name, e0, e1, algo_bytes, flops, tag))
    else:
        code = fn(*args)
        if _PROBE and name in _PROBE:
            fn(*args)
    if code != 0:
        check(code, name)
