"""Builds csrc/librdf_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import hashlib
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


BUILD_ID_MARKER = b"rdf-build-id:"


def source_id():
    """16 hex digits of a SHA-256 over everything the library is built from: the sources, the headers, the compiler flags.
    Baked into the library (rdf_build_id) -- file times say nothing about a .so that travelled with a snapshot."""
    h = hashlib.sha256()
    for p in sorted(SOURCES + HEADERS, key=os.path.basename):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def built_id(path=None):
    """The id baked into a built library, read from the file (no dlopen); None if there is none."""
    try:
        with open(path or SO, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(BUILD_ID_MARKER)
    if at < 0:
        return None
    return blob[at + len(BUILD_ID_MARKER):at + len(BUILD_ID_MARKER) + 16].decode("ascii", "replace")


def sources_present():
    return all(os.path.exists(p) for p in SOURCES + HEADERS)


def is_stale():
    """True when csrc/librdf_hip.so is missing or was built from other sources than the ones next to it."""
    if not os.path.exists(SO):
        return True
    return sources_present() and built_id() != source_id()


def build(force=False, verbose=False):
    """Compile the HIP extension in-tree.  Returns the path of the shared library."""
    if not force and not is_stale():
        return SO
    cmd = [hipcc()] + HIPCC_FLAGS + [f'-DRDF_BUILD_ID="{source_id()}"', "-o", SO + ".tmp"] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
