"""Committed golden vectors (tests/golden/rdf_golden_v1.npz, made by tests/golden/make_golden.py).
CPU: both oracle restatements reproduce them.  GPU: see test_gpu_parity.py."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdf_golden_v1.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_g1_flat_forest(g, impl, oracle, oracle_np):
    mod = oracle if impl == "c" else oracle_np
    out = np.full(g["g1_labels"].shape, 65535, np.uint16)
    mod.eval_forest(g["g1_depth"], g["g1_forest"], out)
    assert np.array_equal(out, g["g1_labels"])


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_g2_reduce_scale_filter(g, impl, oracle, oracle_np):
    mod = oracle if impl == "c" else oracle_np
    out = np.zeros(g["g2_labels"].shape, np.uint16)
    mod.eval_forest(g["g2_depth"], g["g2_forest"], out, 2, g["g2_filter"], 1, 0.5)
    assert np.array_equal(out, g["g2_labels"])


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_g3_layered(g, impl, oracle, oracle_np):
    mod = oracle if impl == "c" else oracle_np
    l0 = np.full(g["g3_l0"].shape, 65535, np.uint16)
    l1, comp = l0.copy(), l0.copy()
    mod.eval_forest(g["g3_depth"], g["g3_forest0"], l0, 2)
    mod.eval_forest(g["g3_depth"], g["g3_forest1"], l1, 2, l0, 3, 1.0)
    assert mod.composite([l0[0], l1[0]], g["g3_cond"], comp) == 0
    assert np.array_equal(l0, g["g3_l0"]) and np.array_equal(l1, g["g3_l1"]) and np.array_equal(comp, g["g3_comp"])


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_g4_single_tree(g, impl, oracle, oracle_np):
    mod = oracle if impl == "c" else oracle_np
    out = np.full(g["g4_labels"].shape, 7, np.uint16)
    mod.eval_tree(g["g4_depth"], g["g4_tree"], out)
    assert np.array_equal(out, g["g4_labels"])
