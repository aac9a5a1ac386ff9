"""Every packed weight of a training step, re-packed by ONE kernel launch (pf_pack_gather_f32, csrc/norm_bwd.hip).

The forward and backward kernels read their weights in layouts of their own -- MFMA operand order, zero padded to the
tile width, and for the data gradients flipped and transposed (train_ops.py).  Built with torch each of them is 3-5
tiny launches (flip, permute, zeros, copy), ~350 per step.  Each layout is an affine gather of the parameter, so a
table of descriptors in device memory -- built once per model: the addresses of parameters do not change across
optimizer steps -- lets one launch at the start of the step refresh all of them, inside the captured hipGraph.

``TrainPacks(model)`` owns the destination tensors.  ``active()`` makes them visible: ``pointflow._cached_pack`` (the
forward wrappers) finds them under its own keys, ``train_ops`` asks ``get(kind, tensor)`` for the backward forms.
"""
import contextlib

import numpy as np
import torch

from . import _lib
from . import pointflow

_F32 = torch.float32
_DIMS, _SRC = 7, 5


def _dtype():
    dt = np.dtype({"names": ["src", "dst", "total", "ndim", "nsrc", "dshape", "dstride", "off", "lim", "sstride", "M",
                             "pad"],
                   "formats": ["<u8", "<u8", "<i8", "<i4", "<i4", ("<i4", _DIMS), ("<i4", _DIMS), ("<i4", _SRC),
                               ("<i4", _SRC), ("<i4", _SRC), ("<i4", (_SRC, _DIMS)), "<i4"],
                   "offsets": [0, 8, 16, 24, 28, 32, 60, 88, 108, 128, 148, 288],
                   "itemsize": int(_lib.load().pf_pack_desc_bytes())})
    return dt


class TrainPacks(object):
    def __init__(self, model):
        self.model = model
        self.dev = next(model.parameters()).device
        self._recs = []               # (src tensor, dst view, dshape, coords)
        self._dst = {}                # key -> destination tensor
        self._srcs = []
        self._params = [(q, q.data_ptr()) for q in model.parameters()]     # (parameter, address at build time)
        self._build()
        dt = _dtype()
        table = np.zeros((len(self._recs),), dtype=dt)
        self.max_total = 1
        for i, (src, dst, dshape, coords) in enumerate(self._recs):
            r = table[i]
            r["src"], r["dst"] = src.data_ptr(), dst.data_ptr()
            total = 1
            for d in dshape:
                total *= d
            r["total"], r["ndim"], r["nsrc"] = total, len(dshape), len(coords)
            r["dshape"][:len(dshape)] = dshape
            r["dstride"][:len(dshape)] = dst.stride()
            for k, (off, lim, sstride, m) in enumerate(coords):
                r["off"][k], r["lim"][k], r["sstride"][k] = off, lim, sstride
                for dim, coef in m.items():
                    r["M"][k][dim] = coef
            self.max_total = max(self.max_total, total)
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(self.dev)
        self.n = len(self._recs)

    # ------------------------------------------------------------------------------------------
    def _add(self, key, src, dst, dshape, coords, view=None):
        """coords: per source dimension (offset, {dst dim: coefficient}); the limits / strides are the source's."""
        src = src.detach()
        assert src.dtype == _F32 and src.is_cuda
        if key is not None:
            self._dst[key] = dst
        v = dst.view(dshape) if view is None else view
        assert tuple(v.shape) == tuple(dshape), (v.shape, dshape)
        cs = []
        for k, (off, m) in enumerate(coords):
            cs.append((int(off), int(src.shape[k]), int(src.stride(k)), m))
        self._recs.append((src, v, tuple(int(d) for d in dshape), cs))
        self._srcs.append((src, src.data_ptr()))

    def _zeros(self, *shape):
        return torch.zeros(shape, dtype=_F32, device=self.dev)

    # the layouts ------------------------------------------------------------------------------
    def _c2w(self, key, W, dgrad=False):
        """pointflow._pack_conv2d_wide of W (Cout, Cin, K, K), or of its flipped transpose (the stride-1 data gradient)."""
        co_w, ci_w, K, _ = W.shape
        cout, cin = (ci_w, co_w) if dgrad else (co_w, ci_w)           # the convolution this pack serves
        a_out, a_in = (1, 0) if dgrad else (0, 1)                      # which W axis holds its output / input channel
        kh = (K - 1, {0: -1}) if dgrad else (0, {0: 1})
        kw = (K - 1, {1: -1}) if dgrad else (0, {1: 1})
        if cout == 8 and cin >= 8:                     # paired rows: [kh'][kw][kq][s][co][j] = w[co][cq kq + j][kh' - s][kw]
            cq = (cin + 3) // 4
            dst = self._zeros(K + 1, K, 4, 16, cq)
            coords = [None] * 4
            coords[a_out] = (0, {4: 1})
            coords[a_in] = (0, {2: cq, 5: 1})
            coords[2] = (K - 1, {0: -1, 3: 1}) if dgrad else (0, {0: 1, 3: -1})
            coords[3] = kw
            self._add(key, W, dst, (K + 1, K, 4, 2, 8, cq), coords, view=dst.view(K + 1, K, 4, 2, 8, cq))
        elif cout <= 16:
            cq = (cin + 3) // 4
            dst = self._zeros(K, K, 4, 16, cq)
            coords = [None] * 4
            coords[a_out] = (0, {3: 1})
            coords[a_in] = (0, {2: cq, 4: 1})
            coords[2], coords[3] = kh, kw
            self._add(key, W, dst, (K, K, 4, 16, cq), coords)
        else:
            dst = self._zeros(K, K, cin // 8, 2, cout, 4)
            coords = [None] * 4
            coords[a_out] = (0, {4: 1})
            coords[a_in] = (0, {2: 8, 3: 4, 5: 1})
            coords[2], coords[3] = kh, kw
            self._add(key, W, dst, (K, K, cin // 8, 2, cout, 4), coords)

    def _c3(self, key, W, flip_t=False, co0=0, cout=None):
        """pointflow.pack_conv3d_weight of W (Cout, Cin, 3,3,3) read as is, or of its flipped transpose; ``co0`` /
        ``cout``: a slice of the served convolution's output channels (the 8 -> 64 data gradient runs as two halves)."""
        Wf = W.detach().reshape(W.shape[0], W.shape[1], 27)
        co_full, cin = (W.shape[1], W.shape[0]) if flip_t else (W.shape[0], W.shape[1])
        cout = co_full if cout is None else cout
        ncp = (cout + 15) // 16 * 16
        dst = self._zeros(cin // 4, 27, 4, ncp)
        view = dst[..., :cout]
        if flip_t:
            coords = [(0, {0: 4, 2: 1}), (co0, {3: 1}), (26, {1: -1})]
        else:
            coords = [(co0, {3: 1}), (0, {0: 4, 2: 1}), (0, {1: 1})]
        self._add(key, Wf, dst, (cin // 4, 27, 4, cout), coords, view=view)

    def _c3p(self, key, W):
        cout, cin = W.shape[:2]
        dst = self._zeros(cin // 4, 36, 4, 16)
        view = dst.view(cin // 4, 3, 4, 3, 4, 2, 8)
        self._add(key, W, dst, (cin // 4, 3, 4, 3, 4, 2, 8),
                  [(0, {6: 1}), (0, {0: 4, 4: 1}), (0, {1: 1}), (0, {2: 1, 5: -1}), (0, {3: 1})], view=view)

    def _c3b(self, key, W, flip_t=False):
        cin = W.shape[0] if flip_t else W.shape[1]
        dst = self._zeros(3, 3, 3, cin // 16, 4, 64, 4)
        if flip_t:
            coords = [(0, {3: 16, 4: 4, 6: 1}), (0, {5: 1}), (2, {0: -1}), (2, {1: -1}), (2, {2: -1})]
        else:
            coords = [(0, {5: 1}), (0, {3: 16, 4: 4, 6: 1}), (0, {0: 1}), (0, {1: 1}), (0, {2: 1})]
        self._add(key, W, dst, (3, 3, 3, cin // 16, 4, 64, 4), coords)

    def _d3b(self, key, W):
        cin, cout = W.shape[:2]
        Wf = W.detach().reshape(cin, cout, 27)
        dst = self._zeros(27, cin // 16, 4, cout, 4)
        self._add(key, Wf, dst, (27, cin // 16, 4, cout, 4), [(0, {1: 16, 2: 4, 4: 1}), (0, {3: 1}), (0, {0: 1})])

    def _wt(self, key, convs):
        K = convs[0].shape[1]
        cout = sum(int(c.shape[0]) for c in convs)
        nc = (cout + 31) // 32 * 32
        dst = self._zeros(K, nc)
        col = 0
        for c in convs:
            w2 = c.detach().reshape(c.shape[0], K)
            n = int(c.shape[0])
            self._add(None, w2, dst, (K, n), [(0, {1: 1}), (0, {0: 1})], view=dst[:, col:col + n])
            col += n
        self._dst[key] = (dst, cout)                # what pointflow.pack_weight_t returns

    def _rows(self, key, mats, n_out):
        """gemm_rows weights: the row-stacked matrices ``mats`` (each (r_i, n_out)) as column chunks of <= 128, each
        zero padded to 32 / 64 / 128: a list of (col0, width, wt (K, nc))."""
        K = sum(int(m.shape[0]) for m in mats)
        chunks, col = [], 0
        while col < n_out:
            width = min(128, n_out - col)
            nc = 32 if width <= 32 else (64 if width <= 64 else 128)
            dst = self._zeros(K, nc)
            row = 0
            for m in mats:
                m2 = m.detach().reshape(m.shape[0], -1)
                r = int(m2.shape[0])
                self._add(None, m2, dst, (r, width), [(0, {0: 1}), (col, {1: 1})], view=dst[row:row + r, :width])
                row += r
            chunks.append((col, width, dst))
            col += width
        self._dst[key] = chunks

    # ------------------------------------------------------------------------------------------
    def _build(self):
        from . import train_ops
        m = self.model
        for tower in (m.coarse_img_conv, m.flow_img_conv):
            for i, (_, _, conv, _) in enumerate(train_ops._tower_blocks(tower)):
                W = conv.weight
                self._c2w(("c2w", id(W)), W)
                if i > 0:
                    if conv.stride[0] == 1:
                        self._c2w(("c2w_dg", id(W)), W, dgrad=True)
                    else:
                        co, ci, k, _ = W.shape
                        ncp = (ci + 15) // 16 * 16
                        dst = self._zeros(co // 4, k * k, 4, ncp)
                        self._add(("d2_dg", id(W)), W.detach().reshape(co, ci, k * k), dst, (co // 4, k * k, 4, ci),
                                  [(0, {0: 4, 2: 1}), (0, {3: 1}), (0, {1: 1})], view=dst[..., :ci])
        vc = m.coarse_vol_conv
        if vc.base_channels == 8 and vc.in_channels == 64:
            self._c3p(("c3p", id(vc.conv0_1.conv.weight)), vc.conv0_1.conv.weight)
            for name in ("conv1_0", "conv2_0", "conv1_1", "conv2_1"):
                W = getattr(vc, name).conv.weight
                self._c3(("c3", id(W)), W)
            for name in ("conv1_1", "conv2_1"):
                W = getattr(vc, name).conv.weight
                self._c3(("c3_dg", id(W)), W, flip_t=True)
            W01 = vc.conv0_1.conv.weight
            for h in (0, 1):
                self._c3(("c3_dg%d" % h, id(W01)), W01, flip_t=True, co0=32 * h, cout=32)
            self._c3b(("c3b", id(vc.conv3_0.conv.weight)), vc.conv3_0.conv.weight)
            self._c3b(("c3b", id(vc.conv3_1.conv.weight)), vc.conv3_1.conv.weight)
            self._c3b(("c3b_dg", id(vc.conv3_1.conv.weight)), vc.conv3_1.conv.weight, flip_t=True)
            self._d3b(("d3b", id(vc.conv4_0.conv.weight)), vc.conv4_0.conv.weight)
            # data gradients of the decoder (a ConvTranspose3d's is the stride-2 convolution with its weight read
            # (Cout', Cin')) and of conv3_0 (the transposed convolution with its weight read (Cin', Cout'))
            self._c3(("c3", id(vc.conv6_0.conv.weight)), vc.conv6_0.conv.weight)
            self._c3(("c3", id(vc.conv5_0.conv.weight)), vc.conv5_0.conv.weight)
            self._c3b(("c3b", id(vc.conv4_0.conv.weight)), vc.conv4_0.conv.weight)
            self._d3b(("d3b", id(vc.conv3_0.conv.weight)), vc.conv3_0.conv.weight)
            W62 = vc.conv6_2.weight
            dst = self._zeros(W62.shape[1], 27)
            self._add(("c1_dg", id(W62)), W62.detach().reshape(1, W62.shape[1], 27), dst, (W62.shape[1], 27),
                      [(0, {}), (0, {0: 1}), (26, {1: -1})])
        for e in m.flow_edge_conv:
            self._wt(("wt", id(e.conv1.weight), id(e.conv2.weight)), [e.conv1.weight, e.conv2.weight])
            self._rows(("rows", id(e.conv1.weight)), [e.conv1.weight, e.conv2.weight], int(e.conv1.weight.shape[1]))
        for blk in m.flow_mlp[0]:
            W = blk.conv.weight
            self._wt(("wt", id(W)), [W])
            self._rows(("rows", id(W)), [W], int(W.shape[1]))

    # ------------------------------------------------------------------------------------------
    def stale(self):
        """True when a source parameter moved (``.to()``, ``param.data = ...``) or the module holds other Parameter
        OBJECTS than at build time (a parameter replaced by setattr keeps the old object's address unchanged, and the
        packs are keyed by object identity): rebuild."""
        current = list(self.model.parameters())
        if len(current) != len(self._params):
            return True
        return any(t is not q or t.data_ptr() != p for q, (t, p) in zip(current, self._params))

    def run(self):
        _lib.call("pf_pack_gather_f32", _lib.ptr(self.table), self.n, self.max_total, _lib.stream(),
                  algo_bytes=8.0 * sum(r[1].numel() for r in self._recs))

    def get(self, kind, tensor):
        return self._dst.get((kind, id(tensor)))

    @contextlib.contextmanager
    def active(self):
        """Packed-weight requests of the forward wrappers (pointflow._cached_pack) resolve to this table's tensors."""
        saved = pointflow._prepacked
        pointflow._prepacked = self._dst
        try:
            yield self
        finally:
            pointflow._prepacked = saved
