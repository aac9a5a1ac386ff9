"""CPU emulation of the bf16x3 tower kernel (csrc/conv2d_wide.hip, conv2d_wide_split_kernel; EXPERIMENT, PF_MATRIX_SPLIT).

Round 5 lost its GPU access before the kernel could run once, so what CAN be checked without hardware is checked here:

  * the host-side weight layout (pointflow._pack_conv2d_wide_split) against the kernel's own address arithmetic -- the
    emulation below reads the packed buffer through the flat offsets the kernel computes
    (((t * 3 + split) * 2 + h) * Cout + co) * 8 + j and the staged patch through (16 kb + 8 h + j), K block by K block,
    exactly as lane (m, h) of the wave does;
  * the arithmetic contract: three round-to-nearest bf16 terms per operand (hi + mid + lo == the float32 value), the six
    products a_h b_l + a_l b_h + a_m b_m + a_h b_m + a_m b_h + a_h b_h with float32 accumulation -- against a float64
    convolution, beside the error of a plain float32 convolution of the same operands (what today's f32-MFMA kernel
    delivers: an fmaf chain).  Not a statement about the hardware's rounding inside a 16-deep MFMA block; that and the
    speed need an MI355X (tools/microbench_split.py).
"""
import pytest
import torch
import torch.nn.functional as F

from pointmvsnet_amd import pointflow

SIX = ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0))       # (split of a, split of b) in the kernel's order, smallest first


def _emulate(x, w, k, stride):
    """y (N, Cout, Ho, Wo) as conv2d_wide_split_kernel computes it (one output pixel = one A row of a 32-row tile)."""
    N, Cin, Hi, Wi = x.shape
    Cout = w.shape[0]
    pad = k // 2
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    KB = Cin // 16
    wp = pointflow._pack_conv2d_wide_split(w)
    flat = wp.reshape(-1).to(torch.float64)                      # the buffer the kernel indexes, bf16 values widened exactly
    xs = [t.to(torch.float64) for t in pointflow.split3_bf16(F.pad(x, (pad, pad, pad, pad)))]   # staged patch: 3 planes
    assert torch.equal((xs[0] + xs[1] + xs[2]).float(), F.pad(x, (pad, pad, pad, pad)))
    y = torch.zeros((N, Cout, Ho, Wo), dtype=torch.float32)
    co = torch.arange(Cout)
    for t in range(k * k * KB):
        tap, kb = divmod(t, KB)
        kh, kw = divmod(tap, k)
        block = torch.zeros((N, Cout, Ho, Wo), dtype=torch.float64)
        for h in (0, 1):
            c0 = 16 * kb + 8 * h
            # A: lane (pixel, h) holds channels c0 .. c0 + 7 of the patch pixel under tap (kh, kw), per split
            a = [p[:, c0:c0 + 8, kh:kh + stride * Ho:stride, kw:kw + stride * Wo:stride] for p in xs]   # (N, 8, Ho, Wo)
            # B: lane (co, h) holds 8 consecutive bf16 at the kernel's flat offset, per split
            b = []
            for sp in range(3):
                off = (((t * 3 + sp) * 2 + h) * Cout + co) * 8                                       # (Cout,)
                b.append(flat[off.unsqueeze(1) + torch.arange(8).unsqueeze(0)])                      # (Cout, 8)
            for sa, sb in SIX:
                block += torch.einsum("njhw,cj->nchw", a[sa], b[sb])
        y = y + block.to(torch.float32)                          # float32 accumulator across the K blocks
    return y


@pytest.mark.parametrize("cin,cout,k,stride,hw", [(64, 64, 3, 1, (6, 20)), (32, 32, 3, 1, (9, 17)),
                                                 (32, 64, 5, 2, (12, 18)), (16, 32, 5, 2, (11, 21))])
def test_split_kernel_layout_and_arithmetic_by_emulation(cin, cout, k, stride, hw):
    g = torch.Generator().manual_seed(cin + 7 * k)
    x = torch.relu(torch.randn((2, cin) + hw, generator=g) * 1.7 + 0.2)      # post-ReLU activations, like the staged patch
    w = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, stride, k // 2)
    y = _emulate(x, w, k, stride)
    f32 = F.conv2d(x, w, None, stride, k // 2)                               # a float32 convolution of the same operands
    scale = float(ref.abs().max())
    e_split = float((y.double() - ref).abs().max()) / scale
    e_f32 = float((f32.double() - ref).abs().max()) / scale
    print("bf16x3 emulation %d->%d k%d/%d: max error / max|y| = %.2e   (float32 convolution: %.2e)"
          % (cin, cout, k, stride, e_split, e_f32))
    # same class as float32 rounding: the three dropped terms are < 2^-24 |a b| each, the accumulation is float32
    assert e_split < 1e-6, (e_split, e_f32)
    assert e_split < 4.0 * e_f32 + 2e-7, (e_split, e_f32)


def test_split3_is_exact_and_ordered():
    g = torch.Generator().manual_seed(3)
    t = torch.randn(100000, generator=g) * torch.exp(4.0 * torch.randn(100000, generator=g))
    hi, mid, lo = pointflow.split3_bf16(t)
    assert torch.equal(hi.float() + mid.float() + lo.float(), t)
    assert bool((mid.float().abs() <= hi.float().abs() * 2.0 ** -8 + 1e-45).all())
    assert bool((lo.float().abs() <= hi.float().abs() * 2.0 ** -16 + 1e-45).all())


def test_split_pack_is_the_documented_layout():
    w = torch.arange(32 * 16 * 9, dtype=torch.float32).reshape(32, 16, 3, 3) / 64.0     # exactly representable in bf16 x 3
    wp = pointflow._pack_conv2d_wide_split(w)
    assert wp.shape == (9, 3, 2, 32, 8) and wp.dtype == torch.bfloat16 and wp.is_contiguous()
    hi, mid, lo = pointflow.split3_bf16(w)
    for (t, sp, h, co, j) in ((0, 0, 0, 0, 0), (4, 1, 1, 17, 5), (8, 2, 0, 31, 7), (5, 0, 1, 3, 2)):
        kh, kw = divmod(t, 3)                                                         # (Cin = 16: one K block per tap)
        want = (hi, mid, lo)[sp][co, 8 * h + j, kh, kw]
        assert wp[t, sp, h, co, j] == want
