"""Stage the reference's own hot-path modules next to the oracle  --  TEST INFRASTRUCTURE, build container only.

    python oracle/make_ref.py           (also called by __graft_entry__.build())

Two things need the reference's OWN code on the GPU box, where ``/root/reference`` does not exist:

* north_star requires that the reference's ``pointmvsnet/model.py`` "consumes the new ops unchanged":
  tests/test_gpu_model.py::test_reference_model_py_runs_unchanged_on_our_operators executes that file on the HIP
  operators (``reference_model_py.txt``, staged since round 2);
* ``bench.py``'s ``cpu_baseline`` should time the reference's CPU path itself (reference test.py:58-69,83-84), not
  a port of it: round 4 stages the modules that path imports -- model.py, networks.py, nn/, functions/*.py (not the
  CUDA extension sources), utils/feature_fetcher.py, utils/torch_utils.py: the files SURVEY.md section 8(a) cites --
  as the package ``oracle/_ref/pointmvsnet/``.

Everything is copied byte for byte from ``/root/reference`` into ``oracle/_ref/`` -- a directory listed in .gitignore
(it never enters the history) but not in .gpurunignore (it travels to the GPU box with the built .so files) -- with a
sha256 manifest.  Nothing under pointmvsnet_amd/ reads it; only tests and bench.py's cpu_baseline leg do.  Where
``/root/reference`` is absent (the GPU box) this is a no-op and whatever was staged at build time is used.
"""
import hashlib
import importlib.machinery
import importlib.util
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
REF = os.path.join(REF_ROOT, "pointmvsnet", "model.py")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "reference_model_py.txt")      # executed by compat.load_reference_model
PKG = os.path.join(OUT_DIR, "pointmvsnet")                 # the hot-path modules as <module path>.py.txt (see activate())
PACKAGE_FILES = ["__init__.py", "model.py", "networks.py",
                 "nn/__init__.py", "nn/conv.py", "nn/mlp.py", "nn/linear.py", "nn/init.py", "nn/functional.py",
                 "functions/__init__.py", "functions/functions.py", "functions/gather_knn.py",
                 "utils/__init__.py", "utils/feature_fetcher.py", "utils/torch_utils.py"]


def reference_root():
    """Where the reference's ``pointmvsnet`` can be imported from: its tree in the build container, the staged copy on
    the GPU box (after ``activate()``), None when neither exists."""
    if os.path.isdir(os.path.join(REF_ROOT, "pointmvsnet")):
        return REF_ROOT
    if os.path.isfile(os.path.join(PKG, "model.py.txt")):
        return OUT_DIR
    return None


class _StagedFinder(object):
    """Import ``pointmvsnet[.x.y]`` from oracle/_ref/pointmvsnet/x/y.py.txt (the staged files keep the ``.txt`` suffix
    of round 2's reference_model_py.txt: they are fixtures, not sources of this repository)."""

    def find_spec(self, name, path=None, target=None):
        if name != "pointmvsnet" and not name.startswith("pointmvsnet."):
            return None
        rel = name.split(".")[1:]
        as_pkg = os.path.join(PKG, *(rel + ["__init__.py.txt"]))
        as_mod = os.path.join(PKG, *rel) + ".py.txt"
        if os.path.isfile(as_pkg):
            loader = importlib.machinery.SourceFileLoader(name, as_pkg)
            return importlib.util.spec_from_file_location(name, as_pkg, loader=loader,
                                                          submodule_search_locations=[os.path.dirname(as_pkg)])
        if rel and os.path.isfile(as_mod):
            loader = importlib.machinery.SourceFileLoader(name, as_mod)
            return importlib.util.spec_from_file_location(name, as_mod, loader=loader)
        return None


def activate():
    """Make ``import pointmvsnet`` resolve to the reference: sys.path for its real tree, a meta-path finder for the
    staged copy.  Returns the root used, or None."""
    root = reference_root()
    if root == REF_ROOT:
        if REF_ROOT not in sys.path:
            sys.path.insert(0, REF_ROOT)
    elif root is not None and not any(isinstance(f, _StagedFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StagedFinder())
    return root


def stage(verbose=True):
    if not os.path.isfile(REF):
        return OUT if os.path.isfile(OUT) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    shutil.copyfile(REF, OUT)
    digest = hashlib.sha256(open(OUT, "rb").read()).hexdigest()
    with open(OUT + ".sha256", "w") as f:
        f.write("%s  %s\n" % (digest, REF))
    lines = []
    for rel in PACKAGE_FILES:
        src = os.path.join(REF_ROOT, "pointmvsnet", rel)
        dst = os.path.join(PKG, rel + ".txt")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        lines.append("%s  pointmvsnet/%s" % (hashlib.sha256(open(dst, "rb").read()).hexdigest(), rel))
    with open(os.path.join(OUT_DIR, "pointmvsnet.sha256"), "w") as f:
        f.write("\n".join(lines) + "\n")
    if verbose:
        print("staged %s -> %s (sha256 %s) and %d package files -> %s" % (REF, OUT, digest[:16], len(PACKAGE_FILES), PKG))
    return OUT


if __name__ == "__main__":
    stage()
