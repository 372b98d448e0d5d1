"""mean-shift modes of the composite label map + fingertip heights.
CPU: the numpy restatement against hand-derived answers.  GPU (-m gpu): rdf_mean_shift /
rdf_fingertip_heights against the restatement within 1e-9 (the reference's own fp64 atomics are
order-dependent, so this row has no bit-exact target), plus run-to-run bitwise reproducibility."""
import importlib

import numpy as np
import pytest

from oracle import mean_shift_numpy as ms_np


def _label_map(seed=3, h=120, w=212, n_classes=6, absent=(5,)):
    rng = np.random.default_rng(seed)
    lab = np.full((h, w), 65535, dtype=np.uint16)
    yy, xx = np.mgrid[0:h, 0:w]
    for c in range(1, n_classes + 1):
        if c in absent:
            continue
        cx, cy = rng.uniform(20, w - 20), rng.uniform(15, h - 15)
        a, b = rng.uniform(4, 14), rng.uniform(4, 14)
        m = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1
        lab[m] = c
        for _ in range(3):   # outliers that the kernel weighting should discount
            lab[rng.integers(0, h), rng.integers(0, w)] = c
    lab[0, 0] = 0
    lab[1, 1] = 60000        # label beyond num_classes: ignored here (the reference would fault)
    return lab


def test_known_answers_numpy():
    # one pixel per class: round 0 puts the mean on it, later rounds shift by exactly 0 (diff = 0, weight 1)
    lab = np.zeros((5, 7), np.uint16)
    lab[2, 3], lab[4, 6] = 1, 2
    m = ms_np.mean_shift(lab, 3, [1.0, 2.0, 3.0], 6)
    assert m[0].tolist() == [3.0, 2.0] and m[1].tolist() == [6.0, 4.0] and np.isnan(m[2]).all()
    # two pixels of a class, symmetric: the centroid is a fixed point (shifts cancel exactly)
    lab = np.zeros((1, 9), np.uint16)
    lab[0, 2] = lab[0, 6] = 1
    m = ms_np.mean_shift(lab, 1, [2.0], 4)
    assert m[0].tolist() == [4.0, 0.0]
    # zero rounds: the zero-initialised means come back (mean_shift.py:33)
    assert (ms_np.mean_shift(lab, 1, [2.0], 0) == 0).all()
    # height: identity plane => -z ; mean (10.9, 5.2) truncates to pixel (10, 5), times labels_reduce 2
    depth = np.arange(40 * 60, dtype=np.uint16).reshape(40, 60)
    hts = ms_np.fingertip_heights(np.array([[10.9, 5.2], [np.nan, 1.0], [100.0, 3.0]]), [1, 2, 3], depth, 2,
                                  400.0, 400.0, 30.0, 20.0, np.eye(4, dtype=np.float32))
    assert hts[0] == -float(depth[10, 20]) and np.isnan(hts[1]) and np.isnan(hts[2])


@pytest.mark.gpu
def test_mean_shift_matches_restatement_and_is_reproducible(rdf, gpu_runtime):
    msmod = importlib.import_module("3d-beats_amd.cuda.mean_shift")
    ms = msmod.MeanShift()
    for seed, (h, w), L in [(3, (120, 212), 6), (4, (240, 424), 7), (5, (33, 65), 2), (6, (480, 848), 6)]:
        lab = _label_map(seed, h, w, L, absent=(min(5, L),))
        var = np.linspace(6.0, 14.0, L).astype(np.float32)
        dl, dv = rdf.to_device(lab[None]), rdf.to_device(var)
        for rounds in (0, 1, 6):
            got = ms.run(rounds, dl, L, dv)
            want = ms_np.mean_shift(lab, L, var, rounds)
            assert got.shape == (L, 2) and got.dtype == np.float64
            assert np.array_equal(np.isnan(got), np.isnan(want))
            ok = ~np.isnan(want)
            assert np.abs(got[ok] - want[ok]).max() < 1e-9, (seed, rounds, np.abs(got[ok] - want[ok]).max())
            again = ms.run(rounds, dl, L, dv)
            assert np.array_equal(got.view(np.uint64), again.view(np.uint64)), "not bitwise reproducible"


@pytest.mark.gpu
def test_fingertip_heights_match_restatement(rdf, gpu_runtime):
    msmod = importlib.import_module("3d-beats_amd.cuda.mean_shift")
    rng = np.random.default_rng(9)
    depth = rdf.synth.frames(["dense"], 77, 480, 848)[0]
    means = np.array([[100.7, 50.2], [423.9, 239.9], [424.0, 10.0], [-0.5, 3.0], [np.nan, 1.0], [12.0, 240.0], [0.0, 0.0]])
    plane = (np.eye(4) + 0.1 * rng.standard_normal((4, 4))).astype(np.float32)
    ids = [1, 2, 3, 4, 5, 6, 7]
    got = msmod.fingertip_heights(rdf.to_device(means), ids, rdf.to_device(depth), 2, 421.3, 420.9, 423.1, 238.6, plane)
    want = ms_np.fingertip_heights(means, ids, depth, 2, 421.3, 420.9, 423.1, 238.6, plane)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], rtol=1e-6, atol=1e-6), (got, want)
    assert np.isnan(want[[2, 4, 5]]).all() and not np.isnan(want[[0, 1, 3, 6]]).any()


@pytest.mark.gpu
def test_mean_shift_class_bigger_than_the_lds_list(rdf, gpu_runtime):
    """A class with more pixels than one workgroup lists in LDS (32 768): the rest is rescanned from the label image every
    round -- same means (1e-9 px), still bitwise reproducible; a neighbouring small class is unaffected."""
    msmod = importlib.import_module("3d-beats_amd.cuda.mean_shift")
    ms = msmod.MeanShift()
    h, w, L = 240, 424, 3
    rng = np.random.default_rng(12)
    lab = np.full((h, w), 1, np.uint16)                 # class 1: ~97 000 pixels
    lab[rng.random((h, w)) < 0.03] = 65535
    lab[100:130, 200:260] = 2                           # class 2: 1 800 pixels; class 3 absent
    lab[0, 0] = 0
    var = np.array([80.0, 9.0, 5.0], np.float32)
    dl, dv = rdf.to_device(lab[None]), rdf.to_device(var)
    assert (lab == 1).sum() > 2 * 32768
    for rounds in (1, 5):
        got = ms.run(rounds, dl, L, dv)
        want = ms_np.mean_shift(lab, L, var, rounds)
        assert np.isnan(got[2]).all() and np.isnan(want[2]).all()
        assert np.abs(got[:2] - want[:2]).max() < 1e-9, np.abs(got[:2] - want[:2]).max()
        assert np.array_equal(got.view(np.uint64), ms.run(rounds, dl, L, dv).view(np.uint64))


@pytest.mark.gpu
def test_mean_shift_and_heights_in_one_launch_equal_the_two_calls(rdf, gpu_runtime):
    """rdf_mean_shift_heights (what HandPipeline uses): the same means and the same heights, bit for bit, as rdf_mean_shift
    followed by rdf_fingertip_heights -- with ids that name no class (0, L + 1), an id whose class has no pixel (NaN mode),
    repeated ids, outputs in device memory and in pinned host memory the kernel writes directly."""
    import torch
    msmod = importlib.import_module("3d-beats_amd.cuda.mean_shift")
    ms = msmod.MeanShift()
    h, w, L, r = 240, 424, 6, 2
    lab = _label_map(23, h, w, L, absent=(4,))
    depth = rdf.synth.frames(["dense"], 91, h * r, w * r)[0]
    var = np.full(L, 10.0, np.float32)
    plane = (np.eye(4) + 0.1 * np.random.default_rng(5).standard_normal((4, 4))).astype(np.float32)
    ids = np.array([1, 6, 4, 0, 7, 2, 2, 5, 3], np.int32)
    intr = (421.3, 420.9, 423.1, 238.6)
    dl, dv, dd = rdf.to_device(lab[None]), rdf.to_device(var), rdf.to_device(depth)
    d_ids, d_plane = rdf.to_device(ids), rdf.to_device(plane)
    for rounds in (1, 6):
        means = ms.run_device(rounds, dl, L, dv)
        want_m = means.get()
        want_h = msmod.fingertip_heights(means, [int(i) for i in ids], dd, r, *intr, plane)
        # device outputs
        out = rdf.DeviceArray((2 * L + len(ids),), np.float64).fill(7.0)
        ms.run_device_with_heights(rounds, dl, L, dv, d_ids, len(ids), dd, r, intr, d_plane, out.ptr, out.ptr + 16 * L)
        got = out.get()
        assert np.array_equal(got[:2 * L].view(np.uint64), want_m.reshape(-1).view(np.uint64))
        assert np.array_equal(got[2 * L:].view(np.uint64), np.asarray(want_h).view(np.uint64)), (got[2 * L:], want_h)
        assert np.isnan(want_h[[2, 3, 4]]).all() and not np.isnan(want_h[[0, 1, 5, 6, 7, 8]]).any()
        # pinned host outputs, written by the kernel
        host = gpu_runtime.alloc_host_mapped((2 * L + len(ids)) * 8)
        assert host is not None
        host[2].view(np.float64)[:] = 7.0
        ms.run_device_with_heights(rounds, dl, L, dv, d_ids, len(ids), dd, r, intr, d_plane, host[1], host[1] + 16 * L)
        torch.cuda.synchronize()
        assert np.array_equal(host[2].view(np.float64).view(np.uint64), got.view(np.uint64))


@pytest.mark.gpu
def test_null_labels_are_refused_not_dereferenced(rdf, gpu_runtime):
    """The one-launch kernel lists every class's pixels whatever the number of rounds: a NULL label image (or variances)
    is an argument error, also with num_rounds = 0."""
    lib = gpu_runtime.lib
    var = rdf.to_device(np.full(4, 10.0, np.float32))
    means = rdf.DeviceArray((4, 2), np.float64)
    lab = rdf.to_device(np.zeros((1, 8, 8), np.uint16))
    assert lib.rdf_mean_shift(None, 8, 8, 4, var.ptr, 0, means.ptr, None, gpu_runtime.stream()) == -2
    assert lib.rdf_mean_shift(lab.ptr, 8, 8, 4, None, 0, means.ptr, None, gpu_runtime.stream()) == -2
    assert lib.rdf_mean_shift(lab.ptr, 8, 8, 4, var.ptr, 0, means.ptr, None, gpu_runtime.stream()) == 0
    assert lib.rdf_mean_shift(None, 0, 8, 4, var.ptr, 0, means.ptr, None, gpu_runtime.stream()) == 0   # no pixel to read
