"""Stage the reference's own model graph next to the oracle  --  TEST INFRASTRUCTURE, build container only.

    python oracle/make_ref.py           (also called by __graft_entry__.build())

north_star requires that the reference's ``pointmvsnet/model.py`` "consumes the new ops unchanged".  The GPU
box has no ``/root/reference``, so the one test that executes that file on the HIP operators
(tests/test_gpu_model.py::test_reference_model_py_runs_unchanged_on_our_operators) used to skip there.  This
script copies that ONE file, byte for byte, from ``/root/reference`` into ``oracle/_ref/`` -- a directory
listed in .gitignore (it never enters the history) but not in .gpurunignore (it travels to the GPU box
with the built .so files) -- together with its sha256.  Nothing under pointmvsnet_amd/ reads it; only
tests do, and only to prove the drop-in claim.  Where ``/root/reference`` is absent (the GPU box) this is a
no-op and whatever was staged at build time is used.
"""
import hashlib
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pointmvsnet/model.py"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "reference_model_py.txt")      # executed by compat.load_reference_model


def stage(verbose=True):
    if not os.path.isfile(REF):
        return OUT if os.path.isfile(OUT) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    shutil.copyfile(REF, OUT)
    digest = hashlib.sha256(open(OUT, "rb").read()).hexdigest()
    with open(OUT + ".sha256", "w") as f:
        f.write("%s  %s\n" % (digest, REF))
    if verbose:
        print("staged %s -> %s (sha256 %s)" % (REF, OUT, digest[:16]))
    return OUT


if __name__ == "__main__":
    stage()
