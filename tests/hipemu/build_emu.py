"""Build libpointflow_emu.so: kernels of pointmvsnet_amd/csrc recompiled for the HOST on top of tests/hipemu's HIP shim.

    python tests/hipemu/build_emu.py            # -> tests/hipemu/build/libpointflow_emu.so

Test infrastructure only (tests/test_emulated_kernels.py).  The kernel sources are used UNCHANGED except for the two
spellings of shared memory, which a host compiler cannot give HIP's meaning: `extern __shared__ ... lds[]` becomes a
pointer to the emulator's one static buffer, `__shared__ T name[...]` a static array (blocks run one after the other).
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pointmvsnet_amd", "csrc")
OUT = os.path.join(HERE, "build")
LIB = os.path.join(OUT, "libpointflow_emu.so")
SOURCES = ["conv2d_wide.hip"]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _host_source(text):
    text = re.sub(r"extern\s+__shared__\s+__attribute__\(\(aligned\(16\)\)\)\s+float\s+lds\[\];",
                  "float* lds = ::hipemu_shared_memory();", text)
    assert "extern __shared__" not in text, "an extern __shared__ declaration the emulator does not know"
    return re.sub(r"\b__shared__\s+", "static ", text)


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "hipemu.cpp"),
                                                       os.path.join(HERE, "hip", "hip_runtime.h"), __file__]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    units = [os.path.join(HERE, "hipemu.cpp")]
    for s in SOURCES:
        dst = os.path.join(OUT, s.replace(".hip", "_host.cpp"))
        with open(os.path.join(CSRC, s)) as f:
            text = _host_source(f.read())
        with open(dst, "w") as f:
            f.write("float* hipemu_shared_memory();\n" + text)
        units.append(dst)
    shared = os.path.join(OUT, "shared_memory.cpp")
    with open(shared, "w") as f:
        f.write("alignas(64) static float g_lds[160 * 1024 / 4];\nfloat* hipemu_shared_memory() { return g_lds; }\n")
    units.append(shared)
    cmd = [CLANG, "-std=c++20", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-w",
           "-I" + HERE, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-x", "c++"] + units + ["-o", LIB]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
