"""tools/roofline.py, the model behind bench.py's `roofline` object, on the counters committed under profiles/ (CPU
only): every level is a share of the kernel's duration in (0, 1], the bound is the largest, and the L1 level follows the
measured curve of tools/ubench_l1_fill.hip (profiles/r02_ubench_l1_fill.txt)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def roofline():
    spec = importlib.util.spec_from_file_location("roofline", os.path.join(ROOT, "tools", "roofline.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_l1_curve_is_the_calibration(roofline):
    f = roofline.l1_cycles_per_access
    assert f(0.0) == pytest.approx(0.592)
    assert f(1.0) == pytest.approx(2.32)
    assert f(0.452) == pytest.approx(1.097)
    # monotone, and between the calibration points linear
    xs = [i / 200 for i in range(201)]
    ys = [f(x) for x in xs]
    assert all(b >= a for a, b in zip(ys, ys[1:]))
    assert f((0.315 + 0.452) / 2) == pytest.approx((0.827 + 1.097) / 2)
    # the additive model of round 1 (0.6 per hit + 2.35 per fill) overstates a half-and-half mix by > 10 %
    assert 0.5 * 0.6 + 0.5 * 2.35 > 1.1 * f(0.5)


def test_committed_counters_give_levels_within_one(roofline):
    line = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
    legs = {"headline": line["roofline"], "cfg2": line["cfg2_single_frame"]["roofline"],
            "cfg5": line["cfg5_shard"]["roofline"]}
    for name, r in legs.items():
        m = roofline.model(dict(r["counters"]), r["kernel_ms"], alg_bytes=r["algorithmic"]["bytes_per_launch"])
        assert set(m["levels"]) == {"hbm", "l2_l1", "l1_ta", "valu"}, name
        for lvl, v in m["levels"].items():
            assert 0.0 < v["frac"] <= 1.0, (name, lvl, v["frac"])
        assert m["frac"] == max(v["frac"] for v in m["levels"].values()), name
        assert m["bound"] in ("valu", "l1_ta"), (name, m["bound"])
        assert m["algorithmic"]["over_hbm_peak"] > 1.0        # the 8(d) figure bounds nothing: kept, not used as frac
        # the recorded object (computed on the GPU box with the profiled passes' own durations) says the same
        assert r["frac"] <= 1.0 and r["bound"] == max(r["levels"], key=lambda k: r["levels"][k]["frac"]), name
        ta = r["levels"]["l1_ta"]
        if ta.get("ta_busy_frac_counter"):
            # the model of the L1 level against the hardware's own busy counter
            assert abs(ta["frac"] - ta["ta_busy_frac_counter"]) < 0.1, (name, ta["frac"], ta["ta_busy_frac_counter"])


def test_model_without_counters_says_so(roofline):
    m = roofline.model(None, 4.4, alg_bytes=1e9)
    assert m["bound"] is None and m["frac"] is None and m["levels"] == {}
    assert m["algorithmic"]["bytes_per_launch"] == 10 ** 9


def test_useful_fraction_never_exceeds_the_utilisation(roofline):
    """`useful_frac` prices the lines the walk needs (rdf_eval_forest_packed_stats), `frac` the accesses the kernel issued
    (hardware counter): issued >= useful, so useful_frac <= frac, on the committed round-4 line and on a made-up launch."""
    c = {"TCP_TCC_READ_REQ_sum": 0.76e9, "TCP_TOTAL_CACHE_ACCESSES_sum": 1.93e9, "SQ_INSTS_VALU": 1.8e9, "GRBM_GUI_ACTIVE": 8 * 8.8e6,
         "_ns:GRBM_GUI_ACTIVE": 3.85e6}
    useful = {"records": 1.0e9, "leaf_rows": 0.02e9, "far_probes": 0.45e9, "blocks": 0}
    m = roofline.model(c, 3.85, useful=useful)
    ta = m["levels"]["l1_ta"]
    assert ta["useful_line_accesses_per_launch"] == int(1.47e9) and 0 < ta["useful_frac"] < ta["frac"] <= 1.0
    assert ta["useful_share_of_issued"] == pytest.approx(1.47 / 1.93, abs=1e-3)
    assert m["useful_frac"] == (ta["useful_frac"] if m["bound"] == "l1_ta" else m["frac"])
    path = os.path.join(ROOT, "profiles", "r04_bench.json")
    if os.path.exists(path):
        line = json.load(open(path))
        for r in (line["roofline"], line["cfg2_balanced"]["batch"]["roofline"], line["cfg5_balanced"]["roofline"]):
            ta = r["levels"]["l1_ta"]
            assert ta["useful_line_accesses_per_launch"] <= ta["l1_line_accesses_per_launch"]
            assert ta["useful_frac"] <= ta["frac"] <= 1.0


def test_ta_busy_model_follows_the_counter_on_round_fives_five_legs(roofline):
    """What TA_TA_BUSY should read = the L1 level's calibrated cycles + a price per L2 miss: for the deep-block kernels a curve over
    the fraction of the fabric's ceiling the launch runs at, measured with the throttled gather microbenchmark
    (profiles/r05_ta_busy_per_miss.txt, nothing fitted); for the heap-order kernels one constant fitted on this file's three such legs.  On the committed round-5 line it is within 10 % of the counter on all five legs -- round 4's
    model, the L1 level alone, read 0.60 and 0.46 against 0.94 and 0.97 on the deep legs -- and recomputing it here from the
    committed counters gives the committed figure."""
    line = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    legs = {"headline": line["roofline"], "cfg2": line["cfg2_single_frame"]["roofline"], "cfg5": line["cfg5_shard"]["roofline"],
            "headline_balanced": line["cfg2_balanced"]["batch"]["roofline"], "cfg5_balanced": line["cfg5_balanced"]["roofline"]}
    for name, r in legs.items():
        ta = r["levels"]["l1_ta"]
        assert ta["ta_busy_frac_counter"] and ta["ta_busy_model"], name
        assert abs(ta["ta_busy_model"] / ta["ta_busy_frac_counter"] - 1.0) < 0.10, (name, ta["ta_busy_model"], ta["ta_busy_frac_counter"])
        assert ta["ta_busy_model"] >= ta["frac"]                    # the L1 level's own cycles are part of it
        deep = roofline.is_deep_kernel(r["kernel"])       # (config 5's "full" forest: the tuner may take the last blocks, by 2 %)
        assert deep or not name.endswith("balanced"), (name, r["kernel"])
        assert not deep or name != "headline", (name, r["kernel"])
        hbm = r["levels"]["hbm"]
        per_miss = roofline.ta_busy_cycles_per_l2_miss("deep" if deep else "divergent", hbm["achieved"] / roofline.GATHER_CEILING_GBS)
        assert per_miss == pytest.approx(ta["ta_busy_cycles_per_l2_miss"], abs=0.02), name
        cyc = ta["peak"] * 1e6
        again = ta["frac"] + r["counters"]["TCC_MISS_sum"] * per_miss / roofline.CUS / cyc
        assert again == pytest.approx(ta["ta_busy_model"], abs=3e-3), name
        # the fabric level: the data-sheet peak (the contract's figure) and the measured ceiling of the walk's own access pattern
        assert hbm["gather_ceiling"] == roofline.GATHER_CEILING_GBS and hbm["frac_of_gather_ceiling"] == pytest.approx(
            hbm["achieved"] / roofline.GATHER_CEILING_GBS, abs=1e-3)
        assert hbm["frac"] < hbm["frac_of_gather_ceiling"] <= 1.0
    # the deep legs are bound by the fabric or within a few per cent of it, and nowhere near 1 on the L1's own line rate
    b, c5 = legs["headline_balanced"], legs["cfg5_balanced"]
    assert c5["bound"] == "hbm" and c5["levels"]["hbm"]["frac_of_gather_ceiling"] > 0.75
    assert b["levels"]["hbm"]["frac"] > 0.6 and b["levels"]["l1_ta"]["frac"] < 0.75


def test_ta_busy_price_per_miss_is_the_calibration(roofline):
    f = roofline.ta_busy_cycles_per_l2_miss
    assert f("deep", 0.68) == pytest.approx(1.94) and f("deep", 0.98) == pytest.approx(6.46) and f("deep", 0.05) == pytest.approx(2.5)
    assert f("deep", 0.83) == pytest.approx(1.94 + (6.46 - 1.94) * 0.5)
    assert f("divergent", 0.3) == 5.6 and f("divergent", 0.8) == 5.6 and f("divergent", 0.99) == pytest.approx(7.74)
    assert f("deep", 1.7) == pytest.approx(6.46) and f("divergent", -1.0) == 5.6


def test_kernel_name_tells_the_deep_walk(roofline):
    assert roofline.is_deep_kernel("void (anonymous namespace)::k_eval_forest<512, true, 4, false, 4, false, 1, false, true>((anonymous namespace)::EvalArgsN<1>)")
    assert not roofline.is_deep_kernel("void (anonymous namespace)::k_eval_forest<512, true, 4, false, 4, false, 1, false, false>((anonymous namespace)::EvalArgsN<1>)")
    assert not roofline.is_deep_kernel("k_eval_forest") and not roofline.is_deep_kernel(None)
