"""The C-ABI library builds for gfx950, loads, and exports every symbol include/rdf_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

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
    assert lib.rdf_forest_packed_bytes(4, 20) == (4 << 20) * 48
    assert b"2^31" in lib.rdf_error_string(-3)


def test_code_object_targets_gfx950(rdf):
    from importlib import import_module
    so = import_module("3d-beats_amd._build").SO
    blob = open(so, "rb").read()
    assert b"gfx950" in blob


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
