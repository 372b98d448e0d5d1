"""The C-ABI library builds for gfx950, loads, and exports every symbol include/rdf_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rdf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_three_reference_kernels():
    names = _declared()
    for n in ("rdf_eval_forest", "rdf_eval_tree", "rdf_composite"):
        assert n in names


def test_library_builds_and_exports_every_declared_symbol(rdf):
    from importlib import import_module
    build = import_module("3d-beats_amd._build")
    so = build.build()
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    for n in _declared():
        assert hasattr(lib, n), f"{n} declared in include/rdf_hip.h but not exported"


def test_binding_table_matches_header(rdf):
    from importlib import import_module
    _lib = import_module("3d-beats_amd._lib")
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.rdf_abi_version() == _lib.ABI_VERSION
    # two tables per heap slot (16-byte hot record, PDF rows); up to four classes: the 64-byte records of the deepest level and a
    # 64-byte trailer; up to eight classes and five levels or more: the deep blocks (128-byte aligned), a zero line and a
    # 128-byte trailer; last: the 128-byte info block
    def deep(T, D, last):       # lines: three-level blocks rooted on levels R0 - 3, R0 - 6, ... >= 0, then the last blocks
        r0 = D - last
        return T * (((1 << r0) - (1 << (r0 % 3))) // 7) + (T << r0) + 2

    def up128(n):
        return (n + 127) & ~127
    assert lib.rdf_forest_packed_bytes(4, 20, 4) == up128((4 << 20) * (16 + 32) + (4 << 19) * 64 + 64) + deep(4, 20, 2) * 128 + 128
    assert lib.rdf_forest_packed_bytes(2, 1, 3) == up128((2 << 1) * (16 + 32)) + 128
    assert lib.rdf_forest_packed_bytes(2, 4, 3) == up128((2 << 4) * (16 + 32) + (2 << 3) * 64 + 64) + 128
    assert lib.rdf_forest_packed_bytes(3, 10, 5) == up128((3 << 10) * (16 + 64)) + deep(3, 10, 1) * 128 + 128
    assert lib.rdf_forest_packed_bytes(3, 10, 9) == up128((3 << 10) * (16 + 96)) + 128
    assert lib.rdf_forest_packed_bytes(0, 10, 4) == 0 and lib.rdf_forest_packed_bytes(4, 0, 4) == 0
    # deep blocks for forests of 5 to 24 levels and up to eight classes, whatever the number of trees
    lib.rdf_forest_packed_bytes.restype, lib.rdf_forest_packed_bytes.argtypes = ctypes.c_size_t, [ctypes.c_int] * 3
    for T in (6, 8):
        assert lib.rdf_forest_packed_bytes(T, 24, 4) == up128((T << 24) * (16 + 32) + (T << 23) * 64 + 64) + deep(T, 24, 2) * 128 + 128
    assert lib.rdf_forest_packed_bytes(2, 25, 4) == up128((2 << 25) * (16 + 32) + (2 << 24) * 64 + 64) + 128
    assert lib.rdf_forest_packed_bytes(8, 22, 4) == up128((8 << 22) * (16 + 32) + (8 << 21) * 64 + 64) + deep(8, 22, 2) * 128 + 128
    assert b"2^31" in lib.rdf_error_string(-3)


def test_code_object_targets_gfx950(rdf):
    from importlib import import_module
    so = import_module("3d-beats_amd._build").SO
    blob = open(so, "rb").read()
    assert b"gfx950" in blob


def test_code_object_stays_small(rdf, tmp_path):
    """Every instantiation of the forest kernel costs compile time and code size; the dispatch in rdf_hip.hip lists the
    ones launches really take.  Fewer than 75 of them (16 walk the deep blocks, 4 count visits and lines), and a library under 1.9 MB."""
    import shutil
    import subprocess
    from importlib import import_module
    so = import_module("3d-beats_amd._build").build()
    assert os.path.getsize(so) < 1_900_000, os.path.getsize(so)
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    local = tmp_path / "lib.so"
    shutil.copy(so, local)
    subprocess.check_call([objdump, "--offloading", str(local)], cwd=tmp_path, stdout=subprocess.DEVNULL)
    kernels = set()
    for f in os.listdir(tmp_path):
        if "gfx950" in f:
            out = subprocess.run([objdump, "-t", str(tmp_path / f)], capture_output=True, text=True).stdout
            kernels |= {l.split()[-1] for l in out.splitlines()
                        if "k_eval_forest" in l and " F " in l and not l.split()[-1].endswith(".kd")}
    assert 0 < len(kernels) < 75, len(kernels)


def test_no_gpu_means_loud_failure_not_cpu_fallback(rdf):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    prev = rdf.set_runtime(None)
    try:
        with pytest.raises(rdf.RdfError):
            rdf.get_runtime()
        with pytest.raises(rdf.RdfError):
            rdf.DecisionTreeEvaluator()
    finally:
        rdf.set_runtime(prev)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "3d-beats_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "librdf_oracle" not in src, f


def test_header_is_plain_c(tmp_path):
    """include/rdf_hip.h is what a C (or cgo / JNI / N-API) binding would include: it must compile as pedantic C99."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "rdf_hip.h"\nint main(void) { return rdf_abi_version() > 0 ? 0 : 1; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(src)])


@pytest.mark.gpu
def test_plain_c_consumer_of_the_shared_library(rdf, gpu_runtime, tmp_path):
    """examples/eval_forest.c, built with gcc against librdf_hip.so and the HIP runtime's C API, evaluates a one-node
    forest and checks the labels itself."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(rdf.library_path())
    exe = tmp_path / "eval_forest_example"
    subprocess.check_call([gcc, "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"),
                           "-I", "/opt/rocm/include", os.path.join(root, "examples", "eval_forest.c"),
                           "-L", libdir, "-l:librdf_hip.so", "-L", "/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "PASS" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_a_library_built_from_other_sources_is_refused(rdf, tmp_path, monkeypatch):
    """Build identity: librdf_hip.so carries a hash of the sources it was built from (rdf_build_id); the binding recomputes it
    from csrc/ + include/ and refuses a mismatch -- a library with today's ABI number and yesterday's kernels passes every
    other check (the .so is git-ignored and travels to the GPU box with the snapshot; file times mean nothing there)."""
    import shutil
    from importlib import import_module
    build = import_module("3d-beats_amd._build")
    _lib = import_module("3d-beats_amd._lib")
    so = build.build()
    assert build.built_id() == build.source_id() and not build.is_stale()
    lib = ctypes.CDLL(so)
    lib.rdf_build_id.restype = ctypes.c_char_p
    assert _lib.check_build_id(lib, so) == build.source_id()
    # the same library with another id baked in (as if built before the last edit of a kernel)
    blob = bytearray(open(so, "rb").read())
    at = blob.find(build.BUILD_ID_MARKER) + len(build.BUILD_ID_MARKER)
    assert at >= len(build.BUILD_ID_MARKER) and blob.count(build.BUILD_ID_MARKER) == 1
    blob[at:at + 16] = b"0123456789abcdef"
    stale = tmp_path / "librdf_hip_stale.so"
    stale.write_bytes(bytes(blob))
    assert build.built_id(str(stale)) == "0123456789abcdef"
    old = ctypes.CDLL(str(stale))
    old.rdf_build_id.restype = ctypes.c_char_p
    with pytest.raises(_lib.RdfError, match="built from other sources"):
        _lib.check_build_id(old, str(stale))
    monkeypatch.setenv("RDF_ALLOW_STALE_LIBRARY", "1")
    with pytest.warns(UserWarning, match="built from other sources"):
        _lib.check_build_id(old, str(stale))
    monkeypatch.delenv("RDF_ALLOW_STALE_LIBRARY")
    # is_stale() sees an edited source without looking at file times
    monkeypatch.setattr(build, "HIPCC_FLAGS", build.HIPCC_FLAGS + ["-DSOMETHING_ELSE"])
    assert build.is_stale()
