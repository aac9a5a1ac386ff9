"""File formats of the evaluation output: PFM maps and DTU camera text files.

Same function names, signatures and BYTES as reference utils/io.py:15-145 (``load_cam_dtu``, ``write_cam_dtu``,
``load_pfm``, ``write_pfm``, ``mkdir``) -- what ``tools/depthfusion.py`` and fusibile read back.  Written from
the format, not from the reference's code; tests/test_eval_output.py compares against files the reference's own
writers produced (tests/golden/make_eval_golden.py).
"""
import os
import re
import sys

import numpy as np


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def pfm_header(width, height, color=False, scale=1.0, little_endian=True):
    """b'Pf\\n<w> <h>\\n<scale>\\n': a negative scale marks little-endian samples."""
    return b"%s\n%d %d\n%f\n" % (b"PF" if color else b"Pf", width, height, -scale if little_endian else scale)


def write_pfm_body(file, body, width, height, scale=1):
    """Header + ``body`` = the samples already in file order (bottom row first, little-endian float32): what the
    device-side packers of csrc/eval_out.hip produce."""
    with open(file, "wb") as f:
        f.write(pfm_header(width, height, False, float(scale), True))
        f.write(body)


def write_pfm(file, image, scale=1):
    image = np.asarray(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(file, "wb") as f:
        f.write(pfm_header(image.shape[1], image.shape[0], color, float(scale), little))
        f.write(np.ascontiguousarray(image[::-1]).tobytes())          # rows bottom-up


def load_pfm(file):
    with open(file, "rb") as f:
        kind = f.readline().rstrip().decode("ascii")
        if kind not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        dims = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("ascii"))
        if not dims:
            raise Exception("Malformed PFM header.")
        width, height = int(dims.group(1)), int(dims.group(2))
        scale = float(f.readline().decode("ascii").rstrip())
        endian = "<" if scale < 0 else ">"
        # np.fromfile like the reference (utils/io.py:96): the caller gets a WRITABLE array -- depthfusion.py's
        # probability_filter masks the loaded depth map in place (tools/depthfusion.py:165-168)
        data = np.fromfile(f, dtype=endian + "f4")
    shape = (height, width, 3) if kind == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def load_cam_dtu(file, num_depth=0, interval_scale=1.0):
    """(2,4,4): [0] extrinsic, [1][:3,:3] intrinsic, [1][3] = (depth_min, interval, num_depth, depth_max)."""
    words = file.read().split()
    cam = np.zeros((2, 4, 4))
    cam[0] = np.array(words[1:17], dtype=np.float64).reshape(4, 4)
    cam[1, :3, :3] = np.array(words[18:27], dtype=np.float64).reshape(3, 3)
    n = len(words)
    if n in (29, 30, 31):
        cam[1, 3, 0] = float(words[27])
        cam[1, 3, 1] = float(words[28]) * interval_scale
        cam[1, 3, 2] = num_depth if n == 29 else float(words[29])
        cam[1, 3, 3] = float(words[30]) if n == 31 else cam[1, 3, 0] + cam[1, 3, 1] * (num_depth - 1)
    return cam


def cam_dtu_text(cam):
    """The text write_cam_dtu writes: every number through str() (NumPy's shortest round-trip repr)."""
    out = ["extrinsic\n"]
    for i in range(4):
        out.append("".join(str(cam[0][i][j]) + " " for j in range(4)) + "\n")
    out.append("\nintrinsic\n")
    for i in range(3):
        out.append("".join(str(cam[1][i][j]) + " " for j in range(3)) + "\n")
    out.append("\n" + " ".join(str(cam[1][3][j]) for j in range(4)) + "\n")
    return "".join(out)


def write_cam_dtu(file, cam):
    with open(file, "w") as f:
        f.write(cam_dtu_text(cam))
