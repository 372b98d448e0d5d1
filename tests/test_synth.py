"""The synthetic workloads (3d-beats_amd/synth.py): the balanced topology -- median thresholds over calibration frames, so
that a forest's deep levels are occupied like a trained forest's -- on the CPU."""
import numpy as np
import pytest


def test_balanced_tree_is_a_complete_tree_with_dyadic_leaves(rdf):
    synth = rdf.synth
    cal = synth.calibration_frames(4, 120, 212)
    t = synth.balanced_tree(3, 9, 4, cal, device="cpu")
    assert t.shape == (511, 15) and t.dtype == np.float32
    assert np.all(t[:255, 5:7] == -1.0) and np.all(t[255:, 5:7] == 0.0)        # every side continues down to level D-1
    assert np.all(t[:255, 7:] == 0.0)
    pdf = t[255:, 7:] * 256.0
    assert np.array_equal(pdf, np.rint(pdf)) and pdf.max() <= 256 and pdf.min() >= 0
    assert np.array_equal(t, synth.balanced_tree(3, 9, 4, cal, device="cpu"))   # seeded: the same tree again
    assert not np.array_equal(t[:, :5], synth.balanced_tree(4, 9, 4, cal, device="cpu")[:, :5])


def test_balanced_forest_occupies_its_deep_levels_and_the_full_topology_does_not(rdf, oracle):
    """What the topology is for: on frames it was NOT calibrated on, a balanced forest's walks spread over
    most of the deepest level; the proposal distribution's own thresholds leave most of it unreachable."""
    synth = rdf.synth
    D = 13
    bal = synth.forest(2, D, 4, "balanced", calib=synth.calibration_frames(8, 240, 424), device="cpu")
    full = synth.forest(2, D, 4, "full")
    frames = synth.mixed_batch(4, 0, 240, 424)
    nb = oracle.distinct_nodes_per_level(frames, bal)
    nf = oracle.distinct_nodes_per_level(frames, full)
    assert nb.shape == (2, D) and np.all(nb[:, 0] == 1)
    assert nb[:, D - 1].min() > 0.8 * (1 << (D - 1)), nb[:, D - 1]
    assert nf[:, D - 1].max() < 0.2 * (1 << (D - 1)), nf[:, D - 1]
    assert np.all(nb[:, :6] == (1 << np.arange(6)))                              # the upper levels completely
    # labels: both restatements agree on the balanced forest too
    want = np.full(frames.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(frames, bal, want, stats=st)
    assert int(st[1]) == int(st[0]) * 2 * D and int(st[2]) == int(st[0]) * 2


def test_balanced_numpy_restatement_agrees(rdf, oracle, oracle_np):
    synth = rdf.synth
    forest = synth.forest(3, 8, 3, "balanced", calib=synth.calibration_frames(2, 60, 100), device="cpu")
    frames = synth.frames(["dense", "live"], 50, 60, 100)
    a = np.full(frames.shape, 65535, np.uint16)
    b = a.copy()
    oracle.eval_forest(frames, forest, a)
    oracle_np.eval_forest(frames, forest, b)
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_balanced_tree_does_not_depend_on_the_device(rdf, gpu_runtime):
    synth = rdf.synth
    cal = synth.calibration_frames(4, 120, 212)
    assert np.array_equal(synth.balanced_tree(1, 11, 4, cal, device="cpu"), synth.balanced_tree(1, 11, 4, cal, device="cuda"))
