"""gather_knn operator.

Same surface as reference ``pointmvsnet/functions/gather_knn.py:10-24``: ``GatherKNN`` is a
``torch.autograd.Function`` over ``(feature (B,C,N), index (B,N,K) int64) -> (B,C,N,K)`` whose backward
returns ``(grad_feature, None)``; ``gather_knn = GatherKNN.apply``.  ``dgcnn_ext`` mirrors the two
functions of the reference's pybind module (``functions/csrc/main.cpp:3-6``) on top of the C ABI
(``pf_gather_knn_{forward,backward}_{f32,f64}``), float and double like the reference dispatch
(``gather_knn_kernel.cu:134``).  Kernels run on the *current* stream and device.  The backward sums every
gradient element in a fixed order over the inverted index lists (``pf_knn_inverse``): unlike the reference's
``atomicAdd`` scatter (``gather_knn_kernel.cu:50-89``) it is bit-reproducible from run to run.
"""
import torch

from .. import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}


def _check(feature_like, index, ndim, what):
    _lib.require_gpu(feature_like, index)
    if feature_like.dim() != ndim or index.dim() != 3:
        raise RuntimeError("%s: expected a %d-d tensor and a 3-d index" % (what, ndim))
    if index.dtype != torch.int64:
        raise RuntimeError("%s: index must be int64" % what)
    if feature_like.dtype not in _SUFFIX:
        raise RuntimeError("%s: only float32 / float64 are supported" % what)
    if index.size(0) != feature_like.size(0) or index.size(1) != feature_like.size(2):
        raise RuntimeError("%s: index shape %s does not match input %s"
                           % (what, tuple(index.shape), tuple(feature_like.shape)))


class _Ext(object):
    """Stand-in for the reference's ``dgcnn_ext`` module (same two function names and argument order)."""

    @staticmethod
    def gather_knn_forward(input, index):
        _check(input, index, 3, "gather_knn_forward")
        B, C, N = input.shape
        K = index.size(2)
        x, idx = input.contiguous(), index.contiguous()
        out = torch.empty((B, C, N, K), dtype=x.dtype, device=x.device)
        with _lib.on_device(x.device):
            _lib.call("pf_gather_knn_forward_" + _SUFFIX[x.dtype], _lib.ptr(x), _lib.ptr(idx), _lib.ptr(out),
                      B, C, N, K, _lib.stream(),
                      algo_bytes=float(B) * (x.element_size() * C * N * (1 + K) + 8.0 * N * K))
        return out

    @staticmethod
    def gather_knn_backward(grad_output, index):
        _check(grad_output, index, 4, "gather_knn_backward")
        B, C, N, K = grad_output.shape
        if index.size(2) != K:
            raise RuntimeError("gather_knn_backward: index.size(2) != grad_output.size(3)")
        g, idx = grad_output.contiguous(), index.contiguous()
        grad_in = torch.empty((B, C, N), dtype=g.dtype, device=g.device)
        with _lib.on_device(g.device):
            # (default: the scatter as a gather over the inverted index lists -- bit-reproducible; the reference's
            # float atomics when pointflow.DETERMINISTIC_BACKWARD is off)
            from .. import pointflow
            det = pointflow.DETERMINISTIC_BACKWARD and B * N * K > 0 and B * N * K < 2 ** 32 - 1 and C <= 65535
            order, start = pointflow.knn_inverse(idx, B, N, K) if det else (None, None)
            _lib.call("pf_gather_knn_backward_" + _SUFFIX[g.dtype], _lib.ptr(g), _lib.ptr(idx), _lib.ptr(grad_in),
                      B, C, N, K, _lib.ptr(order), _lib.ptr(start), _lib.stream(),
                      algo_bytes=float(B) * (g.element_size() * C * N * (1 + K) + 8.0 * N * K))
        return grad_in


dgcnn_ext = _Ext()


class GatherKNN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature, index):
        ctx.save_for_backward(index)
        return dgcnn_ext.gather_knn_forward(feature, index)

    @staticmethod
    def backward(ctx, grad_output):
        (index,) = ctx.saved_tensors
        return dgcnn_ext.gather_knn_backward(grad_output, index), None


gather_knn = GatherKNN.apply
