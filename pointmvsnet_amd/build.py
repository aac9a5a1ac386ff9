"""Build libpointflow_hip.so for gfx950 with hipcc (in-tree, next to this file).

    python -m pointmvsnet_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with gpurun
snapshots.  Flags: -ffp-contract=off keeps the float32 arithmetic exactly as written (the kNN
distance and the bilinear taps must round like the reference's);
-munsafe-fp-atomics selects the hardware global_atomic_add_f32/f64 for the two backward scatters.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libpointflow_hip.so")
SOURCES = ["pf_core.hip", "gather_knn.hip", "knn_lattice.hip", "fetch.hip", "edgeconv.hip", "norm.hip", "conv3d.hip", "conv3d_pair.hip", "deconv3d.hip", "conv3d_bottom.hip", "conv2d.hip", "conv2d_small.hip", "conv2d_wide.hip", "eval_out.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fPIC", "-Wno-pass-failed", "-I" + INCLUDE, "-I" + CSRC]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "pf_common.h"), os.path.join(INCLUDE, "pointflow_hip.h"), __file__]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
