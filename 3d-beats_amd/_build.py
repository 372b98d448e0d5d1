"""Builds csrc/librdf_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rdf_hip.hip")
SOURCES = [SRC, os.path.join(HERE, "csrc", "mean_shift_hip.hip"), os.path.join(HERE, "csrc", "points_ops_hip.hip"),
           os.path.join(HERE, "csrc", "tree_train_hip.hip")]
HEADERS = [os.path.join(HERE, "..", "include", "rdf_hip.h"), os.path.join(HERE, "csrc", "rdf_device.hpp")]
SO = os.path.join(HERE, "csrc", "librdf_hip.so")

# No -ffast-math, no -fgpu-flush-denormals-to-zero: the fp32 divide must stay IEEE-correct
# and denormals must be kept for bit-exact parity (see rdf_hip.hip header).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
               "-fno-fast-math", "-ffp-contract=off"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def is_stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SOURCES + HEADERS + [__file__])


def build(force=False, verbose=False):
    """Compile the HIP extension in-tree.  Returns the path of the shared library."""
    if not force and not is_stale():
        return SO
    cmd = [hipcc()] + HIPCC_FLAGS + ["-o", SO + ".tmp"] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
