"""CPU: the C-ABI library builds for gfx950, loads without a GPU, and exports exactly the symbols
include/pointflow_hip.h declares (and the ctypes prototypes bind the same set)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "pointflow_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", text))


def test_header_library_and_bindings_agree(lib_built):
    from pointmvsnet_amd import _lib
    declared = _header_symbols()
    assert len(declared) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_built]).decode()
    exported = set(re.findall(r" T (pf_[a-z0-9_]+)", out))
    assert declared == exported, (declared ^ exported)
    assert set(_lib.PROTOTYPES) == declared
    lib = _lib.load()
    assert b"gfx950" in lib.pf_version()
    assert b"invalid argument" in lib.pf_error_string(-1)
    assert lib.pf_stat_blocks(1, 25600) == 400 and lib.pf_stat_blocks(16, 96000) == 250
    assert lib.pf_stat_blocks(1, 100) == 2 and lib.pf_stat_blocks(0, 5) == 0


def test_library_contains_gfx950_code_object(lib_built):
    blob = open(lib_built, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))     # the offload bundles' code objects
    assert targets == {b"gfx950"}, targets
    assert b"gfx942" not in blob and b"sm_" not in blob      # single target, no other-arch paths


def test_no_reference_or_oracle_imports_in_product():
    """The product package must not import the oracle (or anything else outside itself + torch)."""
    pkg = os.path.join(ROOT, "pointmvsnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "pointflow_hip.h")).read()
    for cite in ("gather_knn_kernel.cu", "utils/torch_utils.py:16", "utils/feature_fetcher.py:13", "model.py"):
        assert cite in text


def test_bn_job_struct_layout_matches_header(tmp_path):
    """pf_bn_job crosses the ABI by value in an array: the ctypes mirror must have the C compiler's layout."""
    import ctypes
    from pointmvsnet_amd import _lib
    fields = [name for name, _ in _lib.BnJob._fields_]
    src = tmp_path / "layout.c"
    prints = "\n".join('  printf("%%zu\\n", offsetof(pf_bn_job, %s));' % f for f in fields)
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "pointflow_hip.h"\nint main(void) {\n'
                   '  printf("%%zu\\n", sizeof(pf_bn_job));\n%s\n  return 0;\n}\n' % prints)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).decode().split()]
    assert vals[0] == ctypes.sizeof(_lib.BnJob)
    assert vals[1:] == [getattr(_lib.BnJob, f).offset for f in fields]
    # the header is plain C: it compiled above with gcc (no C++-isms at the boundary)


def test_every_tower_layer_runs_on_the_hip_conv_kernel():
    """Every ImageConv layer of the reference widths has exactly one owner, pf_conv2d_wide_f32 (csrc/conv2d_wide.hip);
    other widths fall back to the library convolution + the HIP BatchNorm kernels (never to the CPU)."""
    from pointmvsnet_amd import pointflow
    from pointmvsnet_amd.networks import ImageConv
    tower = ImageConv(8)
    for name in ("conv0", "conv1", "conv2", "conv3"):
        for i, blk in enumerate(getattr(tower, name)):
            conv = blk.conv if hasattr(blk, "conv") else blk
            assert pointflow.conv2d_wide_preferred(conv), (name, i)
    other = ImageConv(12)                                       # widths the kernels are not built for
    assert not pointflow.conv2d_wide_preferred(other.conv1[0].conv)


def test_no_default_path_kernel_uses_scratch_memory():
    """hipcc's per-kernel resource usage, recorded by the build (build/resource_usage.json): a register array that
    the compiler leaves in scratch (e.g. an array of HIP's float4 struct) serialises every access behind a memory
    round trip and is invisible in the source.  No kernel of the library may spill."""
    import json
    import re
    from pointmvsnet_amd import build
    if not os.path.exists(build.USAGE_FILE):
        build.build(verbose=False)
    usage = json.load(open(build.USAGE_FILE))
    assert set(usage) == set(build.SOURCES)
    kernels = 0
    for src, table in usage.items():
        for name, u in table.items():
            kernels += 1
            assert u.get("scratch_bytes_per_lane", 0) == 0, "%s: %s uses %d bytes of scratch per lane" % (
                src, name, u["scratch_bytes_per_lane"])
    assert kernels > 100
