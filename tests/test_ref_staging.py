"""oracle/make_ref.py: the reference's hot-path modules staged byte for byte under the git-ignored oracle/_ref/ (test
infrastructure that travels to the GPU box: bench.py's cpu_baseline times the reference's own CPU path there, and one
GPU test executes the reference's unmodified model.py on the HIP operators)."""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import REFERENCE_DIR, ROOT

sys.path.insert(0, ROOT)
from oracle import make_ref  # noqa: E402


@pytest.mark.skipif(not os.path.isdir(REFERENCE_DIR), reason="build container only: /root/reference is absent")
def test_staged_files_are_the_reference_bytes_and_import_as_the_package():
    make_ref.stage(verbose=False)
    manifest = open(os.path.join(make_ref.OUT_DIR, "pointmvsnet.sha256")).read().split("\n")
    assert len([l for l in manifest if l]) == len(make_ref.PACKAGE_FILES)
    for rel in make_ref.PACKAGE_FILES:
        staged = open(os.path.join(make_ref.PKG, rel + ".txt"), "rb").read()
        assert staged == open(os.path.join(REFERENCE_DIR, "pointmvsnet", rel), "rb").read(), rel
        assert any(l.startswith(hashlib.sha256(staged).hexdigest()) and l.endswith("pointmvsnet/" + rel) for l in manifest)
    # a fresh interpreter that cannot see /root/reference imports the staged package through the meta-path finder
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import make_ref\n"
            "make_ref.REF_ROOT = '/nonexistent'\n"
            "assert make_ref.activate() == make_ref.OUT_DIR\n"
            "import pointmvsnet.model as m, pointmvsnet.utils.torch_utils as t\n"
            "assert m.__file__.endswith('model.py.txt') and hasattr(m, 'PointMVSNet') and hasattr(t, 'get_knn_3d')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]


def test_nothing_staged_is_tracked_by_git():
    out = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout
    assert out.strip() == ""
