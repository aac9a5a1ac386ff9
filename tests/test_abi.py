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
    assert lib.pf_stat_blocks(1, 25600) == 256 and lib.pf_stat_blocks(16, 96000) == 64
    assert lib.pf_stat_blocks(1, 100) == 2 and lib.pf_stat_blocks(0, 5) == 0


def test_library_contains_gfx950_code_object(lib_built):
    blob = open(lib_built, "rb").read()
    assert b"gfx950" in blob
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
