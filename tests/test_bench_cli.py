"""bench.py's contract with the driver, rehearsed on the one GPU of the test box: the N = 1 line carries the roofline and
the legs, and the N > 1 code path (two ranks over gloo sharing the GPU: same control flow as RCCL ranks on distinct
GPUs) gathers, verifies the gather and reports what a driver needs to check the run."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


LINE_LIMIT = 4096      # the driver's record keeps a short tail of stdout: round 4's 26-KB line came back `parsed: null`


def _last_json(stdout):
    """The LAST stdout line: the compact summary the driver parses.  It must be one JSON object under 4 KB."""
    lines = stdout.splitlines()
    assert lines and lines[-1].startswith("{"), stdout[-2000:]
    assert len(lines[-1]) < LINE_LIMIT, len(lines[-1])
    return json.loads(lines[-1])


def _full(compact, path):
    """The full result the compact line points at (`full`): every leg, the roofline levels, the counter dumps."""
    assert compact["full"] == str(path)
    with open(path) as f:
        return json.load(f)


def _check_contract(c):
    """The keys the driver's contract names, on the compact line itself."""
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in c, k
    assert c["unit"] == "Mpix/s" and c["higher_is_better"] is True and c["scaling"] == "weak" and c["vs_baseline"] is None
    assert c["data"] == "synthetic" and c["dtype"] == "f32" and "workload" in c["config"] and "model" not in c["config"]


@pytest.mark.gpu
def test_single_gpu_line_has_roofline_and_legs(tmp_path):
    env = dict(os.environ, RDF_BENCH_CACHE=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--frames", "8",
                        "--depth", "12", "--no-counters", "--no-cfg5", "--cpu-seconds", "1", "--full-json", str(tmp_path / "full.json")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    c = _last_json(r.stdout)
    _check_contract(c)
    # the compact line: a flat roofline, the CPU baseline with its parity verdict, one or two scalars per leg
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes"} <= set(c["roofline"])
    assert all(not isinstance(v, (dict, list)) for v in c["roofline"].values())
    assert c["roofline"]["kernel"].startswith("k_eval_forest") and c["roofline"]["algorithmic_bytes"] > 0
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["differing_pixels"] == 0 and c["cpu_baseline"]["cores"] >= 1
    assert "differ in 0 pixels" in c["cpu_baseline"]["sample"] and c["cpu_baseline"]["value"] > 0
    assert c["value_balanced"] > 0 and c["cfg2_us"] > 0 and c["cfg3_us"] > 0 and c["balanced_differing_pixels"] == 0
    assert "leg_errors" not in c, c.get("leg_errors")
    # the full result is also on stderr (one JSON object), for a reader of the driver's log
    assert any(l.startswith('{"metric"') and len(l) > len(r.stdout.splitlines()[-1]) for l in r.stderr.splitlines())
    d = _full(c, tmp_path / "full.json")
    assert d["value"] == c["value"] and d["ms_per_step"] == c["ms_per_step"]
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "Mpix/s" and d["value"] > 0
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "workload" in d["config"]
    # `value` is the median of the per-step times, the K steps between the barriers are beside it
    # (three steps of a tiny batch: the bracketed wall time is mostly barriers and launch overhead, so the mean rate is only
    # required to exist and not to beat the median by much)
    assert d["value_is"].startswith("median") and d["value_mean"] > 0 and d["value"] > 0.4 * d["value_mean"]
    assert abs(d["ms_per_step"] * d["value"] - 8 * 480 * 848 / 1e3) < 1e-2 * d["ms_per_step"] * d["value"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "levels", "algorithmic"} <= set(d["roofline"])
    assert d["roofline"]["algorithmic"]["bytes_per_launch"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and "differ in 0 pixels" in d["cpu_baseline"]["sample"]
    assert d["cfg2_single_frame"]["kernel_ms"] > 0 and d["cfg3_layered_run"]["ms_per_frame_wall"] > 0
    # round 4: the top level names the bound, the rate per evaluated pixel and the balanced forest's rate; the forest's deep-level
    # table was chosen by DecisionForest.tune; the L1 level carries a useful-work fraction next to its utilisation
    assert "bound" in d and "hbm_frac" in d and 0 < d["value_valid_pixels"] < d["value"] and 0 < d["valid_pixel_share"] < 1
    assert "roofline.bound" in d["metric"] and d["metric"].startswith("classified Mpix/s on 848x480 depth frames (4 trees, depth 20)")
    assert d["config"]["deep_level_table"]["deep_from"] in [int(k) for k in d["config"]["deep_level_table"]["tried"]]
    ta = d["roofline"]["levels"].get("l1_ta")
    if ta:      # (levels come from the committed counters here: --no-counters)
        assert ta["useful_line_accesses_per_launch"] > 0 and ta["useful_frac"] > 0
    b = d["cfg2_balanced"]
    assert "error" not in b, b
    assert b["topology"] == "balanced" and b["parity"]["differing_pixels"] == 0 and b["batch"]["value"] > 0 and d["value_balanced"] == b["batch"]["value"]
    assert len(b["distinct_nodes"]["per_level"]) == 12 and b["distinct_nodes"]["share_of_level"][11] > 2 * b["distinct_nodes_full_topology"]["share_of_level"][11]
    ho = d["cfg3_layered_run"]["host_overhead"]
    assert ho["rdf_layered_run_calls"] == 200 and ho["rdf_eval_forest_packed_calls"] == 200 and ho["rdf_eval_forest_packed_us_in_library_before_the_launch"] < 20


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["p2p", "rccl", "both", "auto", "fastest"])
def test_two_ranks_on_one_gpu_rehearse_the_multi_gpu_line(gather, tmp_path):
    env = dict(os.environ, RDF_BENCH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", "6", "--depth", "12", "--backend", "gloo", "--gather", gather, "--reserve-cus", "0",
           "--cfg5-frames", "2", "--cfg5-trees", "3", "--cfg5-depth", "12", "--full-json", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    c = _last_json(r.stdout)
    _check_contract(c)
    # the compact N > 1 line: which transport `value` is, every mode's figure and verdict, config 5 on all ranks
    assert c["n_gpus"] == 2 and c["config"]["gather_check"] == "ok" and c["distributed"]["rccl_ranks"] == 2
    assert c["distributed"]["distinct_devices"] == 1 and c["distributed"]["kernel_only_ms"] > 0
    assert all(m["gather_check"] == "ok" and m["value"] > 0 for m in c["distributed"]["gather_modes"].values())
    assert c["cfg5_all_ranks"]["gather_check"] == "ok" and c["cfg5_all_ranks"]["direct_stores"]["gather_check"] == "ok"
    if gather == "auto":        # the default: `value` is the RCCL gather's (north_star), the alternatives are beside it
        assert c["config"]["gather"] == "rccl gather" and c["value"] == c["distributed"]["gather_modes"]["rccl gather"]["value"]
    d = _full(c, tmp_path / "full.json")
    assert d["value"] == c["value"] and d["config"]["gather"] == c["config"]["gather"]
    assert d["n_gpus"] == 2 and d["config"]["gather_check"] == "ok"
    dd = d["distributed"]
    assert dd["rccl_ranks"] == 2 and len(dd["devices"]) == 2 and dd["kernel_only_ms"] > 0 and dd["value_kernel_only"] >= d["value"] * 0.5
    assert dd["distinct_devices"] == 1                      # both ranks share the test box's GPU; 8 on the driver's node
    assert all(m["gather_check"] == "ok" for m in dd["gather_modes"].values())
    if gather == "auto":                # (the default: the RCCL gather is `value`)
        assert d["config"]["gather"] == "rccl gather" and d["value"] == dd["gather_modes"]["rccl gather"]["value"]
    elif gather == "fastest":           # (times all three and takes the fastest intact one's figure as `value`)
        best = max(dd["gather_modes"], key=lambda n: dd["gather_modes"][n]["value"])
        assert d["config"]["gather"] == best and d["value"] == dd["gather_modes"][best]["value"]
    else:
        assert ("p2p" in d["config"]["gather"]) == (gather in ("p2p", "both"))
    # both / auto / fastest: copy engines, the kernels' own stores into rank 0's ring, and the RCCL gather
    assert len(dd["gather_modes"]) == (3 if gather in ("both", "auto", "fastest") else 1)
    assert dd["unavailable"] == {}
    if gather in ("p2p", "both", "auto", "fastest"):       # the receive ring's ready counters show the last two steps of both ranks
        assert dd["gather_modes"]["p2p copy engines"]["ready_counters_ok"] is True
    if gather in ("both", "auto", "fastest"):
        assert dd["gather_modes"]["p2p direct stores"]["ready_counters_ok"] is True
    # config 5's workload sharded over the same ranks (1280x720 dense frames; the forest shrunk for the test)
    c5 = d["cfg5_all_ranks"]
    assert c5["n_gpus"] == 2 and c5["gather_check"] == "ok" and c5["value"] > 0 and c5["value_kernel_only"] >= c5["value"]
    # ... and once more with every rank's kernel writing straight into rank 0's ring
    assert c5["p2p_direct_stores"]["gather_check"] == "ok" and c5["p2p_direct_stores"]["value"] > 0


@pytest.mark.gpu
def test_two_ranks_with_split_steps(tmp_path):
    """Round 6: the RCCL mode's steps are split launches (main launch on the CU-masked stream, helper launch behind the previous
    step's gather, one tile queue).  Two ranks on the one GPU, each with its own masked stream and helper stream, gloo standing
    in for RCCL (its wait blocks the host instead of the stream: the ordering is the same): the label maps arrive intact, the
    line says how many workgroups the helper got, and --no-split-step gives the same maps from one launch per step."""
    for extra, split in (([], True), (["--no-split-step"], False)):
        env = dict(os.environ, RDF_BENCH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
               "--frames", "16", "--depth", "12", "--backend", "gloo", "--gather", "rccl", "--reserve-cus", "32", "--no-cfg5",
               "--full-json", str(tmp_path / "full.json")] + extra
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        c = _last_json(r.stdout)
        _check_contract(c)
        m = c["distributed"]["gather_modes"]["rccl gather"]
        assert c["config"]["gather"] == "rccl gather" and c["config"]["gather_check"] == "ok" and m["cus_left_to_rccl"] == 32
        assert c["distributed"]["unavailable"] == {}
        if split:
            assert m["split_step_helper_workgroups"] > 0, m
        else:
            assert "split_step_helper_workgroups" not in m


@pytest.mark.gpu
def test_wall_time_guard_keeps_the_rccl_figure_and_names_what_it_skipped(tmp_path):
    """Round 6: the N>1 default times the RCCL gather -- `value` -- FIRST and always; with the wall-time budget spent (here: a
    budget of zero seconds) the transports beside it and config 5 on all ranks are skipped on every rank alike (rank 0's clock,
    one broadcast per decision) and named in distributed.unavailable; the run still exits 0 with a verified line."""
    env = dict(os.environ, RDF_BENCH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", "6", "--depth", "12", "--backend", "gloo", "--gather", "auto", "--reserve-cus", "0", "--time-budget", "0",
           "--cfg5-frames", "2", "--cfg5-trees", "3", "--cfg5-depth", "12", "--full-json", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    c = _last_json(r.stdout)
    _check_contract(c)
    assert c["n_gpus"] == 2 and c["config"]["gather"] == "rccl gather" and c["config"]["gather_check"] == "ok" and c["value"] > 0
    d = _full(c, tmp_path / "full.json")
    dd = d["distributed"]
    assert list(dd["gather_modes"]) == ["rccl gather"] and dd["gather_modes"]["rccl gather"]["gather_check"] == "ok"
    skipped = dd["unavailable"]
    assert set(skipped) == {"p2p copy engines", "p2p direct stores", "cfg5_all_ranks"}, skipped
    assert all("wall-time budget" in v for v in skipped.values())
    assert "cfg5_all_ranks" not in d


@pytest.mark.gpu
def test_four_ranks_on_one_gpu(tmp_path):
    """The same rehearsal with four ranks (rank-indexed buffers, checksums and the config-5 leg beyond world size 2; four
    processes on the card stay inside the box's limit of six)."""
    env = dict(os.environ, RDF_BENCH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--frames", "3", "--depth", "11", "--backend", "gloo", "--reserve-cus", "0",
           "--cfg5-frames", "1", "--cfg5-trees", "2", "--cfg5-depth", "11", "--full-json", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    c = _last_json(r.stdout)
    _check_contract(c)
    assert c["n_gpus"] == 4 and c["config"]["gather"] == "rccl gather" and c["config"]["gather_check"] == "ok"     # the default
    assert c["distributed"]["device_indices"] == [0, 0, 0, 0] and len(c["distributed"]["gather_modes"]) == 3
    d = _full(c, tmp_path / "full.json")
    assert d["n_gpus"] == 4 and d["config"]["gather_check"] == "ok"
    dd = d["distributed"]
    assert dd["rccl_ranks"] == 4 and len(dd["devices"]) == 4 and sorted(x["rank"] for x in dd["devices"]) == [0, 1, 2, 3]
    assert all(m["gather_check"] == "ok" for m in dd["gather_modes"].values()) and len(dd["gather_modes"]) == 3
    assert dd["unavailable"] == {} and dd["gather_modes"]["p2p copy engines"]["ready_counters_ok"] is True
    assert dd["gather_modes"]["p2p direct stores"]["ready_counters_ok"] is True
    c5 = d["cfg5_all_ranks"]
    assert c5["n_gpus"] == 4 and c5["gather_check"] == "ok" and "4 x 1 dense" in c5["workload"]
    assert c5["p2p_direct_stores"]["gather_check"] == "ok"


@pytest.mark.gpu
def test_a_rank_that_cannot_map_the_ring_turns_the_run_to_the_rccl_gather(tmp_path):
    """What the first real 8-GPU run may meet: one rank's hipIpcOpenMemHandle fails.  bench.py itself (not only the class)
    must agree on every rank, say on the line which leg was unavailable and why, time the RCCL gather instead and exit 0."""
    env = dict(os.environ, RDF_BENCH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", "6", "--depth", "12", "--backend", "gloo", "--gather", "auto", "--reserve-cus", "0", "--no-cfg5",
           "--fail-ipc-open-on-rank", "1", "--full-json", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    c = _last_json(r.stdout)
    assert c["config"]["gather"] == "rccl gather" and c["distributed"]["unavailable"]["p2p"].startswith("unavailable: rank 1: hipIpcOpenMemHandle")
    d = _full(c, tmp_path / "full.json")
    dd = d["distributed"]
    assert d["n_gpus"] == 2 and d["config"]["gather"] == "rccl gather" and d["config"]["gather_check"] == "ok" and d["value"] > 0
    assert list(dd["gather_modes"]) == ["rccl gather"]
    assert dd["unavailable"]["p2p"].startswith("unavailable: rank 1: hipIpcOpenMemHandle")


@pytest.mark.gpu
def test_the_rccl_path_runs_on_one_rank(tmp_path):
    """The N>1 branch of bench.py under the REAL backend (nccl = RCCL) with one rank (--force-distributed): communicator
    init with a device id, the CU-masked compute stream, dist.gather on the side stream, the peer ring (rank 0 is its own
    peer), the kernels' direct stores into it, checksum verification, the `distributed` block and config 5 on all ranks.  It
    proves nothing about xGMI -- the box has one GPU and RCCL refuses two ranks on one device -- but no line of the path the
    driver's 8-GPU run takes is executed there for the first time."""
    env = dict(os.environ, RDF_BENCH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--frames", "8", "--depth", "12", "--backend", "nccl", "--gather", "auto", "--force-distributed",
           "--cfg5-frames", "2", "--cfg5-trees", "3", "--cfg5-depth", "12", "--full-json", str(tmp_path / "full.json")]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    c = _last_json(r.stdout)
    _check_contract(c)
    # the default at N > 1: `value` is the RCCL gather's, on the CU-masked stream
    assert c["config"]["gather"] == "rccl gather" and c["config"]["cus_left_to_rccl"] == 32 and c["distributed"]["backend"] == "nccl"
    assert c["value"] == c["distributed"]["gather_modes"]["rccl gather"]["value"]
    d = _full(c, tmp_path / "full.json")
    dd = d["distributed"]
    assert d["n_gpus"] == 1 and dd["backend"] == "nccl" and dd["rccl_ranks"] == 1 and dd["distinct_devices"] == 1
    assert set(dd["gather_modes"]) == {"p2p copy engines", "p2p direct stores", "rccl gather"}
    assert all(m["gather_check"] == "ok" for m in dd["gather_modes"].values()), dd["gather_modes"]
    assert dd["gather_modes"]["rccl gather"]["cus_left_to_rccl"] == 32          # the CU-masked stream was had
    assert dd["unavailable"] == {} and d["config"]["gather_check"] == "ok" and d["value"] > 0
    c5 = d["cfg5_all_ranks"]
    assert c5["gather_check"] == "ok" and c5["p2p_direct_stores"]["gather_check"] == "ok"
    assert d["distributed"].get("total_seconds", 0) <= time.time() - t0 + 1


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compact_line_of_a_full_result_stays_under_four_kilobytes():
    """No GPU needed: round 4's own full result (26 KB, the line the driver could not parse) and an 8-rank variant of it with
    every optional block present and over-long strings go through bench.compact_line -- contract keys, a flat roofline and the CPU
    baseline survive, the line stays under the limit."""
    bench = _bench_module()
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    c = bench.compact_line(full, "/somewhere/bench_full.json")
    text = json.dumps(c)
    assert len(text) < LINE_LIMIT and json.loads(text) == c
    _check_contract(c)
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["full"] == "/somewhere/bench_full.json"
    r = c["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    assert r["bound"] == full["roofline"]["bound"] and r["frac"] == full["roofline"]["frac"] and r["traffic"] == full["roofline"]["traffic"]
    assert r["hbm_frac"] == full["roofline"]["levels"]["hbm"]["frac"] and r["algorithmic_bytes"] == full["roofline"]["algorithmic"]["bytes_per_launch"]
    assert r["kernel"] == "k_eval_forest<512,true,4,false,4,false,1,false,false>"
    assert c["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and c["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert c["cfg5_balanced_mpix"] == full["cfg5_balanced"]["value"] and c["cfg2_us"] == round(full["cfg2_single_frame"]["kernel_ms"] * 1e3, 1)
    # eight ranks, every optional block, hostile string lengths
    full["n_gpus"] = 8
    full["config"]["workload"] = "w" * 5000
    full["config"]["gather"], full["config"]["gather_check"] = "rccl gather", "ok"
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["distributed"] = {"backend": "nccl", "rccl_ranks": 8, "distinct_devices": 8, "kernel_only_ms": 3.9, "value_kernel_only": 1e5,
                           "devices": [{"rank": g, "local_rank": g, "device_index": g, "uuid": "u" * 40, "name": "AMD Instinct MI355X"} for g in range(8)],
                           "gather_modes": {n: {"ms_per_step": 4.1, "value": 1e5, "cus_left_to_rccl": 32, "gather_check": "ok", "ready_counters_ok": True}
                                            for n in ("p2p copy engines", "p2p direct stores", "rccl gather")},
                           "unavailable": {"p2p": "x" * 3000, "cu_masked_stream": "y" * 3000}, "total_seconds": 99.0}
    full["cfg5_all_ranks"] = {"value": 1e4, "unit": "Mpix/s", "ms_per_step": 10.0, "value_kernel_only": 1.1e4, "kernel_only_ms": 9.0, "n_gpus": 8,
                              "gather": "rccl gather to rank 0 inside the timed region", "gather_check": "ok", "workload": "z" * 900,
                              "p2p_direct_stores": {"value": 1e4, "ms_per_step": 10.0, "gather_check": "ok"}}
    full["some_leg"] = {"error": "e" * 4000}
    c8 = bench.compact_line(full, "/somewhere/bench_full.json")
    assert len(json.dumps(c8)) < LINE_LIMIT
    _check_contract(c8)
    assert c8["distributed"]["rccl_ranks"] == 8 and c8["distributed"]["device_indices"] == list(range(8))
    assert set(c8["distributed"]["gather_modes"]) == {"p2p copy engines", "p2p direct stores", "rccl gather"}
    assert c8["roofline"]["frac"] == full["roofline"]["frac"] and c8["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    # round 6's line, from round 6's own full result: SURVEY 8(d)'s formula under its own name beside the measured HBM fraction,
    # config 5's keys say which topology they are, the trainer's forest has a bound, the library's build id, split steps at N > 1
    with open(os.path.join(ROOT, "profiles", "r06_bench.json")) as f:
        full6 = json.load(f)
    c6 = bench.compact_line(full6, "x.json")
    assert len(json.dumps(c6)) < LINE_LIMIT
    _check_contract(c6)
    assert c6["roofline"]["survey_8d_frac"] == full6["roofline"]["algorithmic"]["over_hbm_peak"] > 1 > c6["roofline"]["hbm_frac"]
    assert "algorithmic_over_hbm_peak" not in c6["roofline"]
    assert c6["cfg5_full_mpix"] == full6["cfg5_shard"]["value"] and c6["cfg5_full_hbm_frac"] < c6["cfg5_balanced_hbm_frac"]
    assert not any(k.startswith("cfg5_") and not k.startswith(("cfg5_full_", "cfg5_balanced_")) for k in c6)
    assert c6["trainer_forest_bound"] == full6["cfg2_trainer_forest"]["roofline"]["bound"] and 0 < c6["trainer_forest_frac"] < 1
    assert len(c6["config"]["library_build_id"]) == 16
    full6["n_gpus"] = 8
    full6["distributed"] = dict(full["distributed"])
    full6["distributed"]["gather_modes"] = {"rccl gather": {"ms_per_step": 4.0, "value": 1e5, "cus_left_to_rccl": 32, "gather_check": "ok",
                                                            "split_step_helper_workgroups": 96}}
    c68 = bench.compact_line(full6, "x.json")
    assert len(json.dumps(c68)) < LINE_LIMIT and c68["distributed"]["gather_modes"]["rccl gather"]["split_step_helper_workgroups"] == 96
