"""The measurement tools and job scripts are not exercised by the GPU suite: at least keep them syntactically alive and
free of references to knobs / entry points that no longer exist."""
import glob
import os
import py_compile
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_compile():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"),
                                                                      os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        py_compile.compile(f, doraise=True)


def test_job_scripts_parse():
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "jobs", "*.sh"))):
        subprocess.check_call(["bash", "-n", f])


def test_tools_name_only_existing_entry_points_and_knobs():
    from pointmvsnet_amd import _lib
    header = open(os.path.join(ROOT, "include", "pointflow_hip.h")).read()
    declared = set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", header))
    product = ""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pointmvsnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                product += open(os.path.join(dirpath, f)).read()
    product += open(os.path.join(ROOT, "bench.py")).read()
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))):
        src = open(f).read()
        for name in set(re.findall(r"\"(pf_[a-z0-9_]+_f(?:32|64))\"", src)):
            assert name in declared or name in _lib.PROTOTYPES, (f, name)
        for knob in set(re.findall(r"\b(PF_[A-Z0-9_]+)\b", src)):
            assert knob in product, "%s mentions %s, which nothing in the product reads" % (os.path.basename(f), knob)
