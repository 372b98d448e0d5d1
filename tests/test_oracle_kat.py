"""Pins the CPU oracle (C and numpy restatements) to the hand-derived known answers of kat_cases.py
and to each other.  CPU only."""
import numpy as np
import pytest

import kat_cases

CASES = kat_cases.cases()
COMPOSITE = kat_cases.composite_cases()


def _run_case(mod, c, prefill):
    if c["kind"] == "forest":
        n, h, w = c["depth"].shape
        r = c["labels_reduce"]
        out = np.full((n, h // r, w // r), prefill, dtype=np.uint16)
        mod.eval_forest(c["depth"], c["forest"], out, r, c["filter"], c["filter_class"], c["scale_factor"])
    else:
        out = np.full(c["depth"].shape, prefill, dtype=np.uint16)
        mod.eval_tree(c["depth"], c["tree"], out)
    return out


@pytest.mark.parametrize("impl", ["c", "numpy"])
@pytest.mark.parametrize("prefill", [65535, 0, 77])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_known_answers(case, prefill, impl, oracle, oracle_np):
    mod = oracle if impl == "c" else oracle_np
    got = _run_case(mod, case, prefill)
    want = kat_cases.expected_array(case["expected"], prefill)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{case['name']} ({case['cite']}):\n got {got}\nwant {want}"


@pytest.mark.parametrize("impl", ["c", "numpy"])
@pytest.mark.parametrize("case", COMPOSITE, ids=[c["name"] for c in COMPOSITE])
def test_composite_known_answers(case, impl, oracle, oracle_np):
    mod = oracle if impl == "c" else oracle_np
    for prefill in (65535, 0):
        h, w = case["images"][0].shape
        out = np.full((1, h, w), prefill, dtype=np.uint16)
        bad = mod.composite(case["images"], case["cond"], out)
        want = kat_cases.expected_array(case["expected"], prefill)
        assert np.array_equal(out, want), f"{case['name']} ({case['cite']})"
        assert bad == case["bad"]


def test_single_tree_forest_equals_tree_kernel_when_leaf_reached(oracle, rdf):
    """with a leaf reached on every walk, a T=1 forest and the tree kernel agree."""
    synth = rdf.synth
    f = synth.forest(1, 6, 4, "trained")
    d = synth.frames(["dense", "live"], 3, 40, 56)
    a = np.full(d.shape, 65535, np.uint16)
    b = a.copy()
    oracle.eval_forest(d, f, a)
    oracle.eval_tree(d, f[0], b)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("topology", ["full", "trained"])
@pytest.mark.parametrize("r,s", [(1, 1.0), (2, 0.5), (3, 2.0)])
def test_c_and_numpy_restatements_agree(topology, r, s, oracle, oracle_np, rdf):
    synth = rdf.synth
    f = synth.forest(3, 7, 5, topology)
    d = synth.frames(["dense", "live", "live"], 10, 45, 61)
    filt = (np.arange(3 * (45 // r) * (61 // r)).reshape(3, 45 // r, 61 // r) % 3).astype(np.uint16)
    for use_filter in (False, True):
        a = np.full((3, 45 // r, 61 // r), 65535, np.uint16)
        b = a.copy()
        sa, sb = np.zeros(3, np.uint64), np.zeros(3, np.uint64)
        kw = dict(filter_images=filt, filter_class=1) if use_filter else {}
        oracle.eval_forest(d, f, a, r, scale_factor=s, stats=sa, **kw)
        oracle_np.eval_forest(d, f, b, r, scale_factor=s, stats=sb, **kw)
        assert np.array_equal(a, b)
        assert np.array_equal(sa, sb)
        assert sa[0] > 0
    t1, t2 = np.full(d.shape, 9, np.uint16), np.full(d.shape, 9, np.uint16)
    oracle.eval_tree(d, f[1], t1)
    oracle_np.eval_tree(d, f[1], t2)
    assert np.array_equal(t1, t2)


def test_oracle_thread_count_does_not_change_results(oracle, rdf):
    synth = rdf.synth
    f = synth.forest(2, 8, 3, "trained")
    d = synth.frames(["dense"], 1, 64, 96)
    a = np.full(d.shape, 65535, np.uint16)
    b = a.copy()
    oracle.eval_forest(d, f, a, n_threads=1)
    oracle.eval_forest(d, f, b, n_threads=0)
    assert np.array_equal(a, b)


def test_full_topology_visit_counts(oracle, rdf):
    """Full topology: every evaluated (pixel, tree) reads D records and reaches one leaf => the
    worst-case algorithmic bytes of SURVEY 8(d): 4 + T(32 D + 4 C) bytes per pixel."""
    synth = rdf.synth
    T, D, C = 4, 6, 4
    f = synth.forest(T, D, C, "full")
    d = synth.frames(["dense"], 0, 32, 48)
    out = np.full(d.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(d, f, out, stats=st)
    npx = 32 * 48
    assert tuple(st) == (npx, npx * T * D, npx * T)
    assert synth.algorithmic_bytes(1, 32, 48, 1, False, C, st) == npx * (4 + T * (32 * D + 4 * C))
    assert oracle.order_sensitive(d, f) == 0  # dyadic PDFs: every summation order gives the same argmax
