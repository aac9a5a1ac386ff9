"""Build libpointflow_hip.so for gfx950 with hipcc (in-tree, next to this file).

    python -m pointmvsnet_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with gpurun
snapshots.  Flags: -ffp-contract=off keeps the float32 arithmetic exactly as written (the kNN
distance and the bilinear taps must round like the reference's);
-munsafe-fp-atomics selects the hardware global_atomic_add_f32/f64 for the two backward scatters.
"""
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libpointflow_hip.so")
SOURCES = ["pf_core.hip", "gather_knn.hip", "knn_lattice.hip", "fetch.hip", "edgeconv.hip", "norm.hip", "conv3d.hip", "conv3d_pair.hip", "deconv3d.hip", "conv3d_bottom.hip", "conv2d_wide.hip", "eval_out.hip", "knn_inverse.hip", "norm_bwd.hip", "conv_wgrad.hip", "conv_dgrad.hip", "warp_bwd.hip", "train_heads.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fPIC", "-Wno-pass-failed", "-I" + INCLUDE, "-I" + CSRC]
# every compile also reports the per-kernel resource usage; it is kept next to the object as JSON and merged into
# build/resource_usage.json (tests/test_abi.py: no kernel of the default path may touch scratch memory)
USAGE_FLAG = "-Rpass-analysis=kernel-resource-usage"
USAGE_FILE = os.path.join(HERE, "build", "resource_usage.json")
_FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
           "Occupancy [waves/SIMD]": "waves_per_simd", "LDS Size [bytes/block]": "static_lds_bytes"}


def _parse_usage(stderr_text):
    """{mangled kernel name: {vgprs, agprs, scratch_bytes_per_lane, ...}} from hipcc's resource-usage remarks."""
    out, cur = {}, None
    for line in stderr_text.splitlines():
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: \s*([A-Za-z][A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in _FIELDS:
            cur[_FIELDS[m.group(1).strip()]] = int(m.group(2))
    return out


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
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INCLUDE, "pointflow_hip.h"), __file__]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        u = o.replace(".o", ".usage.json")
        if force or _stale(o, [s] + headers) or not os.path.exists(u):
            cmd = [hipcc] + FLAGS + [USAGE_FLAG, "-fno-caret-diagnostics", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            proc = subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True)
            rest = [ln for ln in proc.stderr.splitlines()
                    if "remark:" not in ln and ln.strip() and not ln.startswith("In file included from")]
            if rest:
                sys.stderr.write("\n".join(rest) + "\n")
            if proc.returncode != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
            with open(u, "w") as f:
                json.dump(_parse_usage(proc.stderr), f, indent=1, sort_keys=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    merged = {}
    for src in SOURCES:
        with open(os.path.join(objdir, src.replace(".hip", ".usage.json"))) as f:
            merged[src] = json.load(f)
    with open(USAGE_FILE, "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
