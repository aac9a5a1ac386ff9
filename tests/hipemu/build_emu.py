"""Build libpointflow_emu.so: kernels of pointmvsnet_amd/csrc recompiled for the HOST on top of tests/hipemu's HIP shim.

    python tests/hipemu/build_emu.py            # -> tests/hipemu/build/libpointflow_emu.so

Test infrastructure only (tests/test_emulated_kernels.py).  The kernel sources are used UNCHANGED except for the two
spellings of shared memory, which a host compiler cannot give HIP's meaning: `extern __shared__ ... lds[]` becomes a
pointer to the emulator's per-OS-thread buffer, `__shared__ T name[...]` a `static thread_local` array (a block's GPU
threads are fibers of one OS thread).
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pointmvsnet_amd", "csrc")
ASAN = os.environ.get("HIPEMU_ASAN") == "1"      # AddressSanitizer build: run python under LD_PRELOAD=asan_runtime() (README.md)
OUT = os.path.join(HERE, "build_asan" if ASAN else "build")
LIB = os.path.join(OUT, "libpointflow_emu.so")
SOURCES = ["pf_core.hip", "gather_knn.hip", "knn_lattice.hip", "fetch.hip", "edgeconv.hip", "norm.hip", "conv3d.hip", "conv3d_pair.hip", "deconv3d.hip", "conv3d_bottom.hip", "conv2d_wide.hip", "eval_out.hip", "knn_inverse.hip", "norm_bwd.hip", "conv_wgrad.hip", "conv_dgrad.hip", "warp_bwd.hip", "train_heads.hip"]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _host_source(text):
    text = re.sub(r"extern\s+__shared__\s+__attribute__\(\(aligned\(16\)\)\)\s+(float|char|unsigned char)\s+(\w+)\[\];",
                  lambda m: "%s* %s = reinterpret_cast<%s*>(::hipemu_shared_memory());" % (m.group(1), m.group(2), m.group(1)),
                  text)
    assert "extern __shared__" not in text, "an extern __shared__ declaration the emulator does not know"
    # the two gfx950 instructions written as inline assembly (knn_lattice.hip: keys are never NaN, so these are plain
    # ordered comparisons of doubles)
    text = text.replace('asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));', "r = b < a ? b : a;")
    text = text.replace('asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));', "r = b < a ? a : b;")
    assert "asm(" not in text and "asm volatile" not in text, "inline assembly the emulator does not know"
    return re.sub(r"\b__shared__\s+", "static thread_local ", text)


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
    san = ["-fsanitize=address", "-shared-libasan", "-g", "-fno-omit-frame-pointer"] if ASAN else []
    cmd = [CLANG, "-std=c++20", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-w"] + san + [
           "-I" + HERE, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-x", "c++"] + units + ["-o", LIB]
    subprocess.check_call(cmd)
    return LIB


def asan_runtime():
    return subprocess.check_output([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
