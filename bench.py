#!/usr/bin/env python3
"""bench.py -- classified Mpix/s of the RDF inference path on MI355X.

One "step" = one pass of the forest kernel over this rank's batch of synthetic 848x480 depth frames
(4 trees, depth 20, 4 classes -- BASELINE.json's metric config), frames already resident in HBM.
Default workload: 128 frames per GPU per step (= config 4's shard, 1024 frames / 8 GPUs; it is
config 2's frame x 128, half dense / half live-like), weak scaling: N GPUs evaluate N x 128 frames
and every rank's label maps reach rank 0 over xGMI inside the timed region: `value` is the RCCL gather's
(torch.distributed.gather, overlapped with the next step's launch), two other transports are timed beside it and
reported (copy-engine peer copies into rank 0's IPC-mapped ring; the kernels' own stores into it): DESIGN.md section 6.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0's LAST stdout line is ONE compact JSON object (< 4 KB: the contract's keys, a flat `roofline`, `cpu_baseline`, one
or two scalars per leg); the FULL result -- everything below -- goes to `bench_full.json` in the current directory
(--full-json PATH) and, as one line, to stderr.  At N = 1 the same run also measures
  * `cfg2_single_frame` (ONE 848x480 frame per launch), `cfg3_layered_run` (LayeredDecisionForest.run),
  * `cfg5_shard` (config 5's per-GPU shard: 32 dense 1280x720 frames, 8 trees of depth 22, two frames checked against
    the oracle inside the run),
  * `roofline` objects for the headline kernel, config 2 and config 5: a hierarchical roofline (tools/roofline.py) built
    from rocprofv3 counters that THIS run collects -- before it touches the GPU itself it starts itself once per counter
    set as `rocprofv3 --pmc ... -- python3 bench.py --leg <name>` (separate passes, never combined with tracing),
  * (N > 1 only) `cfg5_all_ranks`: config 5's workload sharded over the same ranks (32 dense 1280x720 frames and the
    replicated T8/D22 forest per rank: at N = 8 this is BASELINE configs[4] itself), timed without and with an RCCL
    gather of the label maps to rank 0, barrier + synchronize on both sides, MAX over ranks, gather verified by checksums,
  * `cpu_baseline`: this repo's CPU restatement (oracle/rdf_oracle.c, OpenMP) -- the reference has no CPU path --
    timed on a bounded sample of the same frames, which also checks the GPU labels of those frames bit for bit.
"""
import argparse
import contextlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CACHE = os.environ.get("RDF_BENCH_CACHE", "/tmp/rdf_bench_cache")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=128, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=848)
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--topology", default="full", choices=["full", "trained", "balanced", "trainer"],
                    help="synthetic topologies; `trainer` = the forest the cfg2_trainer_forest leg trained and left in the bench cache")
    ap.add_argument("--chunks", type=int, default=0, help="eval/gather pipeline chunks per step (0: one launch per step; at N>1 the gather then overlaps the NEXT step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true", help="do not run the rocprofv3 --pmc child passes (roofline levels "
                    "then come from the committed profiles/r0N_roofline_counters.json and say so)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the config-5 shard leg (2 GiB forest)")
    ap.add_argument("--cfg5-frames", type=int, default=32)
    ap.add_argument("--cfg5-trees", type=int, default=8, help="(tests shrink the config-5 forest)")
    ap.add_argument("--cfg5-depth", type=int, default=22)
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer variant (pinned H2D + kernel + D2H)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the legs beside the forest kernel (reference-layout "
                    "forest, trained-like topology, per-hand pipeline, mean shift, training)")
    ap.add_argument("--headline-only", action="store_true", help="only the timed batch (no other legs, no counters): "
                    "what tools/profile.sh traces so that rocprofv3's per-kernel average is the headline kernel's")
    ap.add_argument("--no-balanced", action="store_true", help="skip the legs on forests whose deep levels are occupied "
                    "(cfg2_balanced, cfg5_balanced)")
    ap.add_argument("--no-tune", action="store_true", help="do not let DecisionForest.tune choose the deep-level table per forest")
    ap.add_argument("--deep-from", type=int, default=None, help="the headline forest's deep-level table, fixed instead of tuned: 0 = heap-order "
                    "records, N = deep blocks from level N (what tools/profile.sh passes, so that a kernel trace holds the timed launches only)")
    ap.add_argument("--leg", default=None, choices=["headline", "cfg2", "cfg5", "headline_balanced", "cfg5_balanced", "headline_trainer"],
                    help="internal: run ONE leg and print its kernel time (the program the --pmc passes profile)")
    ap.add_argument("--unpacked", action="store_true", help="evaluate straight from the reference-layout forest")
    ap.add_argument("--scheduler", default="dynamic", choices=["dynamic", "static", "tile"],
                    help="tile schedule of the forest kernel: persistent workgroups on a device-side queue (default), "
                         "persistent with static striding, or one workgroup per tile (non-persistent: 8 %% slower alone, "
                         "but a concurrent RCCL kernel never waits for a free slot)")
    ap.add_argument("--gather", default="auto", choices=["auto", "p2p", "rccl", "both", "fastest"],
                    help="how the label maps reach rank 0 at N>1: rccl = torch.distributed.gather (RCCL over xGMI) on a side stream, "
                         "next to a compute stream that leaves 32 CUs to RCCL's kernels; p2p = every rank copies its shard into "
                         "rank 0's IPC-mapped ring with the copy engines (no CU involved); "
                         "auto (default) = `value` is the RCCL gather's (BASELINE's north_star: a single RCCL gather of the label "
                         "maps), and the two alternatives -- copy engines, and `p2p direct stores` (every rank's kernel writes its "
                         "labels straight into rank 0's ring, no copy) -- are timed back to back beside it and reported in "
                         "`distributed.gather_modes` (the RCCL gather alone if the ring cannot be mapped); fastest = the same three, "
                         "`value` is the fastest intact one's; both = the same, `value` is p2p's")
    ap.add_argument("--reserve-cus", type=int, default=-1, help="run the forest kernel on a stream that leaves this many CUs "
                    "(one per shader engine = 32) to RCCL's kernels; -1: 32 with the rccl gather, 0 otherwise (DESIGN.md section 6)")
    ap.add_argument("--no-split-step", action="store_true", help="RCCL gather mode: one launch per step on the CU-masked stream, "
                    "without the helper launch that takes the reserved CUs back once the previous step's gather has finished")
    ap.add_argument("--time-budget", type=float, default=300.0, help="N>1: seconds of wall time (rank 0's clock, from process "
                    "start) after which the OPTIONAL parts are skipped -- the transports timed beside the RCCL gather, config 5 on "
                    "all ranks -- and named in distributed.unavailable; the RCCL gather's figure (`value`) is always measured")
    ap.add_argument("--fail-ipc-open-on-rank", type=int, default=None, help="test hook: that rank behaves as if it could not map "
                    "rank 0's receive buffer (the run must fall back to the RCCL gather and say why)")
    ap.add_argument("--force-distributed", action="store_true", help="take the N>1 code path (communicator, gather modes, CU-masked "
                    "stream, checksums, `distributed` block, config 5 on all ranks) even with ONE rank: puts the RCCL path on a one-GPU box")
    ap.add_argument("--full-json", default=None, help="where rank 0 writes the FULL result (every leg, counter dumps, censuses); "
                    "default: bench_full.json in the current directory.  The last stdout line is the compact summary of it (< 4 KB)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend at N>1 (nccl = RCCL; gloo only to "
                    "rehearse the control flow with several ranks on one GPU)")
    return ap.parse_args()


class Events:
    """hipEvent pairs recorded on the stream the kernels are launched on (through the C ABI)."""

    def __init__(self, rt, n):
        import ctypes
        self.rt, self.lib, self.ct = rt, rt.lib, ctypes
        self.ev = []
        for _ in range(n):
            e = ctypes.c_void_p()
            assert self.lib.rdf_event_create(ctypes.byref(e)) == 0
            self.ev.append(e)

    def record(self, i):
        assert self.lib.rdf_event_record(self.ev[i], self.rt.stream()) == 0

    def elapsed_ms(self, i, j):
        ms = self.ct.c_float()
        assert self.lib.rdf_event_synchronize(self.ev[j]) == 0
        assert self.lib.rdf_event_elapsed_ms(self.ev[i], self.ev[j], self.ct.byref(ms)) == 0
        return float(ms.value)

    def destroy(self):
        for e in self.ev:
            self.lib.rdf_event_destroy(e)


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cached(name, make):
    """Synthetic inputs are seeded and deterministic; the --pmc child passes reuse what the first process generated."""
    path = os.path.join(CACHE, name + ".npy")
    try:
        return np.load(path, mmap_mode="r")
    except Exception:
        pass
    arr = make()
    try:
        os.makedirs(CACHE, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp.npy"
        np.save(tmp, arr)
        os.replace(tmp, path)
    except Exception:
        pass
    return arr


COMMITTED_COUNTERS = ("r06_roofline_counters.json", "r05_roofline_counters.json", "r04_roofline_counters.json")     # newest first


def committed_counters(key):
    for name in COMMITTED_COUNTERS:
        try:
            got = json.load(open(os.path.join(ROOT, "profiles", name))).get(key)
        except Exception:
            continue
        if got:
            return dict(got, _file=f"profiles/{name}")
    return None


def collect_counters(a, legs):
    """Separate `rocprofv3 --pmc` passes over `python3 bench.py --leg X` (this very file), one process per pass, started
    before this process initialises HIP.  Returns {leg: {"counters":…, "kernel":…, "kernel_ms_profiled":…, "source":…}}."""
    import roofline
    out = {}
    shape = ["--frames", str(a.frames), "--height", str(a.height), "--width", str(a.width), "--trees", str(a.trees),
             "--depth", str(a.depth), "--classes", str(a.classes), "--topology", a.topology, "--scheduler", a.scheduler,
             "--cfg5-frames", str(a.cfg5_frames), "--cfg5-trees", str(a.cfg5_trees), "--cfg5-depth", str(a.cfg5_depth)] \
        + (["--unpacked"] if a.unpacked else [])
    for leg in legs:
        cmd = [sys.executable, os.path.abspath(__file__), "--leg", leg, "--steps", "3", "--warmup", "1"] + shape \
            + (["--no-tune"] if a.no_tune else [])
        t0 = time.perf_counter()
        # (the last launches of a pass are the timed ones: a leg first lets DecisionForest.tune try the other tables; cfg2 times 50)
        name, vals, log = roofline.collect(cmd, "k_eval_forest", timeout=300, passes=roofline.PASSES, last_n=50 if leg == "cfg2" else 3)
        out[leg] = {"counters": vals, "kernel": name, "log": log, "seconds": round(time.perf_counter() - t0, 1),
                    "source": "rocprofv3 --pmc child passes of this run" if vals else None}
    return out


def roofline_for(leg, key, live, kernel_ms, alg_bytes, useful=None):
    import roofline
    c = (live or {}).get(leg, {})
    vals, src, kern = c.get("counters"), c.get("source"), c.get("kernel")
    if not vals:
        com = committed_counters(key)
        if com:
            vals, kern = com.get("counters"), com.get("kernel")
            src = f"{com.get('_file')} (committed; this run collected none)"
    r = roofline.model(vals, kernel_ms, alg_bytes, useful=useful, kernel_name=kern)
    r["kernel"] = kern or "k_eval_forest"
    r["counters_source"] = src
    r["counters"] = {k: (int(v) if v == int(v) else round(v, 1)) for k, v in (vals or {}).items() if not k.startswith("_")}
    return r


LINE_LIMIT = 4000       # bytes of the last stdout line (the driver's record keeps a short tail: round 4's 26-KB line came back unparsed)


def _short_kernel(name):
    """`void (anonymous namespace)::k_eval_forest<512, true, ...>(...)` -> `k_eval_forest<512,true,...>`"""
    if not name:
        return name
    s = str(name).replace("void ", "").replace("(anonymous namespace)::", "")
    s = s.split(">(")[0] + (">" if ">(" in s else "")
    return s.replace(", ", ",")[:96]


def compact_line(out, full_path=None):
    """The ONE line the driver parses: the contract's keys, a flat `roofline`, `cpu_baseline` and one or two scalars per leg.
    Everything else (levels, counter dumps, censuses, tune tables, prose) is in the full result (`full`, and on stderr)."""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d

    def us(ms):
        return None if ms is None else round(ms * 1e3, 1)

    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = "classified Mpix/s on 848x480 depth frames (4 trees, depth 20); % of the bounding roofline level (roofline.bound; HBM: roofline.hbm_frac)"
    for k in ("value_is", "value_mean", "ms_per_step_mean", "ms_per_step_median", "value_valid_pixels", "valid_pixel_share"):
        if k in out:
            line[k] = out[k]
    cfg = out.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload"))[:260]}
    for k in ("frames_per_gpu", "frame", "trees", "tree_depth", "classes", "topology", "forest_layout", "tile_schedule",
              "pipeline_chunks", "sharding", "gather", "gather_overlap", "gather_check", "cus_left_to_rccl", "library_build_id"):
        if cfg.get(k) is not None:
            line["config"][k] = cfg[k]
    line["config"]["deep_from"] = g(cfg, "deep_level_table", "deep_from")

    def flat_roofline(r):
        if not isinstance(r, dict):
            return None
        lv = r.get("levels") or {}
        f = {"bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
             "frac": r.get("frac"), "useful_frac": r.get("useful_frac"), "traffic": r.get("traffic"),
             "kernel": _short_kernel(r.get("kernel")), "kernel_ms": r.get("kernel_ms"),
             "hbm_frac": g(lv, "hbm", "frac"), "hbm_gbs": g(lv, "hbm", "achieved"), "hbm_peak_gbs": g(lv, "hbm", "peak"),
             "l2_l1_frac": g(lv, "l2_l1", "frac"), "l1_ta_frac": g(lv, "l1_ta", "frac"),
             "ta_busy_frac_counter": g(lv, "l1_ta", "ta_busy_frac_counter"), "ta_busy_model": g(lv, "l1_ta", "ta_busy_model"),
             "hbm_frac_of_gather_ceiling": g(lv, "hbm", "frac_of_gather_ceiling"), "valu_frac": g(lv, "valu", "frac"),
             "l2_hit_rate": r.get("l2_hit_rate"), "clock_ghz": r.get("clock_ghz"),
             "algorithmic_bytes": g(r, "algorithmic", "bytes_per_launch"), "algorithmic_gbs": g(r, "algorithmic", "rate_gbs"),
             # SURVEY 8(d)'s figure as written: algorithmic bytes / kernel time / 8 TB/s.  Above 1 on this workload -- the bytes
             # are mostly served by LDS, L1 and L2 -- so it is NOT the fraction of any roofline; hbm_frac is the measured one
             "survey_8d_frac": g(r, "algorithmic", "over_hbm_peak"),
             "copy_ceiling_gbs": g(r, "copy_ceiling", "value"),
             "counters": ("live --pmc passes of this run" if "child passes" in str(r.get("counters_source"))
                          else (str(r.get("counters_source"))[:80] if r.get("counters_source") else None))}
        return {k: v for k, v in f.items() if v is not None or k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}

    line["roofline"] = flat_roofline(out.get("roofline"))
    line["bound"], line["hbm_frac"] = out.get("bound"), out.get("hbm_frac")
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: (str(cb[k])[:300] if k == "sample" else cb[k])
                                for k in ("value", "unit", "cores", "kind", "sample", "parity_frames", "differing_pixels") if k in cb}
        if cb.get("value"):
            line["speedup_vs_cpu_baseline"] = round(out["value"] / cb["value"], 1)
    # ---- one or two scalars per leg ----
    legs = {
        "value_balanced": out.get("value_balanced"),
        "balanced_ms": g(out, "cfg2_balanced", "batch", "ms_per_step"),
        "balanced_bound": g(out, "cfg2_balanced", "batch", "roofline", "bound"),
        "balanced_frac": g(out, "cfg2_balanced", "batch", "roofline", "frac"),
        "balanced_hbm_frac": g(out, "cfg2_balanced", "batch", "roofline", "levels", "hbm", "frac"),
        "balanced_ta_busy": g(out, "cfg2_balanced", "batch", "roofline", "levels", "l1_ta", "ta_busy_frac_counter"),
        "balanced_deep_from": g(out, "cfg2_balanced", "tune", "deep_from"),
        "balanced_differing_pixels": g(out, "cfg2_balanced", "parity", "differing_pixels"),
        "cfg2_us": us(g(out, "cfg2_single_frame", "kernel_ms")),
        "cfg2_frac": g(out, "cfg2_single_frame", "roofline", "frac"),
        "cfg2_balanced_us": us(g(out, "cfg2_balanced", "kernel_ms")),
        "cfg2_trained_us": us(g(out, "cfg2_trained", "kernel_ms")),
        "cfg3_us": us(g(out, "cfg3_layered_run", "ms_per_frame_wall")),
        # config 5's shard twice: `cfg5_full_*` on SURVEY 8(d)'s synthetic "full" forest (walks bunch up: cache-resident, NOT the
        # HBM-bound point), `cfg5_balanced_*` on a forest whose deep levels are occupied (3.6 GiB of tables: HBM-resident)
        "cfg5_full_mpix": g(out, "cfg5_shard", "value"), "cfg5_full_ms": g(out, "cfg5_shard", "kernel_ms"),
        "cfg5_full_bound": g(out, "cfg5_shard", "roofline", "bound"), "cfg5_full_frac": g(out, "cfg5_shard", "roofline", "frac"),
        "cfg5_full_hbm_frac": g(out, "cfg5_shard", "roofline", "levels", "hbm", "frac"),
        "cfg5_full_differing_pixels": g(out, "cfg5_shard", "parity", "differing_pixels"),
        "cfg5_balanced_mpix": g(out, "cfg5_balanced", "value"), "cfg5_balanced_ms": g(out, "cfg5_balanced", "kernel_ms"),
        "cfg5_balanced_bound": g(out, "cfg5_balanced", "roofline", "bound"),
        "cfg5_balanced_frac": g(out, "cfg5_balanced", "roofline", "frac"),
        "cfg5_balanced_hbm_frac": g(out, "cfg5_balanced", "roofline", "levels", "hbm", "frac"),
        "cfg5_balanced_ta_busy": g(out, "cfg5_balanced", "roofline", "levels", "l1_ta", "ta_busy_frac_counter"),
        "cfg5_balanced_frac_of_gather_ceiling": g(out, "cfg5_balanced", "roofline", "levels", "hbm", "frac_of_gather_ceiling"),
        "balanced_frac_of_gather_ceiling": g(out, "cfg2_balanced", "batch", "roofline", "levels", "hbm", "frac_of_gather_ceiling"),
        "cfg5_balanced_deep_from": g(out, "cfg5_balanced", "tune", "deep_from"),
        "cfg5_balanced_differing_pixels": g(out, "cfg5_balanced", "parity", "differing_pixels"),
        "pcie_inclusive_mpix": g(out, "pcie_inclusive", "value"),
        "pcie_pipelined_mpix": g(out, "pcie_inclusive_pipelined", "value"),
        "unpacked_mpix": g(out, "unpacked", "value"),
        "trained_batch_mpix": g(out, "cfg2_trained", "batch", "value"),
        "trainer_forest_mpix": g(out, "cfg2_trainer_forest", "value"),
        "trainer_forest_bound": g(out, "cfg2_trainer_forest", "roofline", "bound"),
        "trainer_forest_frac": g(out, "cfg2_trainer_forest", "roofline", "frac"),
        "trainer_forest_hbm_frac": g(out, "cfg2_trainer_forest", "roofline", "levels", "hbm", "frac"),
        "hand_pipeline_us": g(out, "hand_pipeline", "us_per_hand_per_frame_as_hipgraph"),
        "train_seconds": g(out, "train", "seconds"),
        "cpu_baseline_numpy_mpix": g(out, "cpu_baseline_numpy", "value"),
    }
    line.update({k: v for k, v in legs.items() if v is not None})
    errors = {k: str(v["error"])[:120] for k, v in out.items() if isinstance(v, dict) and "error" in v}
    if errors:
        line["leg_errors"] = errors
    dd = out.get("distributed")
    if isinstance(dd, dict):
        line["distributed"] = {
            "backend": dd.get("backend"), "rccl_ranks": dd.get("rccl_ranks"), "distinct_devices": dd.get("distinct_devices"),
            "device_indices": [d_.get("device_index") for d_ in dd.get("devices") or []],
            "kernel_only_ms": dd.get("kernel_only_ms"), "value_kernel_only": dd.get("value_kernel_only"),
            "gather_modes": {n: {k: m.get(k) for k in ("ms_per_step", "value", "gather_check", "cus_left_to_rccl", "split_step_helper_workgroups") if k in m}
                             for n, m in (dd.get("gather_modes") or {}).items()},
            "unavailable": {k: str(v)[:160] for k, v in (dd.get("unavailable") or {}).items()},
            "total_seconds": dd.get("total_seconds")}
    c5n = out.get("cfg5_all_ranks")
    if isinstance(c5n, dict):
        line["cfg5_all_ranks"] = {k: (str(c5n[k])[:160] if isinstance(c5n[k], str) else c5n[k])
                                  for k in ("value", "unit", "ms_per_step", "value_kernel_only", "kernel_only_ms", "n_gpus", "gather",
                                            "gather_check", "skipped", "error") if k in c5n}
        ds = c5n.get("p2p_direct_stores")
        if isinstance(ds, dict):
            line["cfg5_all_ranks"]["direct_stores"] = {k: (str(v)[:120] if isinstance(v, str) else v) for k, v in ds.items()
                                                       if k in ("value", "ms_per_step", "gather_check", "unavailable", "error")}
    line["full"] = full_path
    # a guarantee, not a hope: optional groups go first if the line is still too long
    for drop in (None, "leg_errors", "cfg5_all_ranks", "legs", "distributed.unavailable", "cpu_baseline.sample", "config.workload"):
        if drop == "legs":
            for k in legs:
                if k not in ("value_balanced", "cfg2_us", "cfg3_us", "cfg5_full_mpix", "cfg5_balanced_mpix", "cfg5_balanced_hbm_frac"):
                    line.pop(k, None)
        elif drop and "." in drop:
            a_, b_ = drop.split(".")
            if isinstance(line.get(a_), dict) and b_ in line[a_]:
                line[a_][b_] = str(line[a_][b_])[:60] if isinstance(line[a_][b_], str) else {}
        elif drop:
            line.pop(drop, None)
        if len(json.dumps(line)) < LINE_LIMIT:
            break
    return line


def emit(out, a):
    """Rank 0: the full result to a side file and to stderr, its compact summary as the LAST stdout line."""
    path = a.full_json or os.path.join(os.getcwd(), "bench_full.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f)
    except OSError as e:
        print(f"could not write {path}: {e}", file=sys.stderr)
        path = None
    print(json.dumps(out), file=sys.stderr, flush=True)
    text = json.dumps(compact_line(out, path))
    assert len(text) < LINE_LIMIT + 96, len(text)
    print(text, flush=True)


def main():
    t_main = time.perf_counter()
    a = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL, peer-mapped buffers); before HIP starts
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    multi = world > 1 or a.force_distributed        # the distributed code path (one rank too, with --force-distributed)
    full_run = not multi and a.leg is None and not a.headline_only

    # ---- counters first: the child passes must run before this process touches the GPU ----
    os.environ.setdefault("RDF_BENCH_RUN_ID", f"{os.getpid()}_{int(time.time())}")      # (inherited by the child passes: tune_forest)
    live = None
    if full_run and not a.no_counters:
        legs = ["headline", "cfg2"] + ([] if a.no_cfg5 else ["cfg5"]) + \
               ([] if a.no_balanced else ["headline_balanced"] + ([] if a.no_cfg5 else ["cfg5_balanced"]))
        try:
            live = collect_counters(a, legs)
        except Exception as e:   # never let the profiler take the measurement down
            print(f"counter collection failed: {type(e).__name__}: {e}", file=sys.stderr)

    import torch
    import torch.distributed as dist

    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N (N>1) must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    rdf = importlib.import_module("3d-beats_amd")
    dmod = importlib.import_module("3d-beats_amd.distributed")
    synth = rdf.synth
    rt = rdf.get_runtime()  # raises without the HIP library or a device: no CPU fallback
    lib = rt.lib
    lib.rdf_set_scheduler({"dynamic": 1, "static": 0, "tile": 2}[a.scheduler])
    ev = rdf.DecisionTreeEvaluator(use_packed=not a.unpacked)
    ev.auto_tune = not a.no_tune       # (every forest of this run is tuned explicitly, outside the timed regions: tune_forest)

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # ================================================================================================================
    # config 5's per-GPU shard: dense 1280x720 frames, T8/D22/C4 full forest (512 MB of hot records: beyond the 256-MB
    # Infinity Cache), one launch per step; two frames compared with the oracle
    # ================================================================================================================
    def tune_forest(forest_obj, sample, key=None):
        """Outside every timed region, like packing: which table serves the forest's deep levels (DecisionForest.tune).
        `key`: ONE choice per forest and bench run -- the --pmc child passes of a leg and this process must walk the same
        table, or a leg's counters describe another kernel than the one it timed (two tables within 2 % of each other, config 5's
        "full" forest, tuned differently from process to process): the first process of the run that tunes a key leaves its
        choice in the bench cache, the others take it."""
        if a.unpacked or a.no_tune:
            return None

        def fixed(level, how):
            assert lib.rdf_forest_set_deep_from(forest_obj.packed(1.0).ptr, int(level)) == 0
            res = {"deep_from": int(level), "tried": None, "chosen_by": how}
            forest_obj.__dict__.setdefault("_tuned", {})[1.0] = res     # (no auto-tune on top)
            return res
        if a.deep_from is not None and a.headline_only:
            return fixed(a.deep_from, "--deep-from")
        run_id = os.environ.get("RDF_BENCH_RUN_ID")
        path = os.path.join(CACHE, f"tune_{run_id}_{key}.json") if (run_id and key and not multi) else None
        if path and os.path.exists(path):
            try:
                got = json.load(open(path))
                return dict(fixed(got["deep_from"], "an earlier process of this bench run"), tried=got.get("tried"))
            except Exception:       # noqa: BLE001 -- (a half-written file: tune here)
                pass
        res = forest_obj.tune(sample)
        if path:
            try:
                os.makedirs(CACHE, exist_ok=True)
                tmp = f"{path}.{os.getpid()}.tmp"
                json.dump(res, open(tmp, "w"))
                os.replace(tmp, path)
            except OSError:
                pass
        return res

    def useful_lines(forest_obj, depth_arr):
        """The lines the timed launch needs from the L1 at the least (rdf_eval_forest_packed_stats: the same launch, same
        geometry and table choice, with counters on) -- the useful-work numerator of the roofline's L1 level."""
        if a.unpacked or int(forest_obj.num_classes) > 4:
            return None, None
        n_, h_, w_ = (int(v) for v in depth_arr.shape)
        st8 = rdf.DeviceArray((8,), np.uint64).fill(0)
        tmp = rdf.DeviceArray((n_, h_, w_), np.uint16).fill(65535)
        rc_ = lib.rdf_eval_forest_packed_stats(depth_arr.ptr, n_, w_, h_, forest_obj.packed(1.0).ptr, forest_obj.forest_cu.ptr,
                                               int(forest_obj.num_trees), int(forest_obj.max_depth), int(forest_obj.num_classes),
                                               tmp.ptr, 1, st8.ptr, rt.stream())
        if rc_ != 0:
            return None, None
        v = [int(x) for x in st8.get()]
        del tmp
        return {"records": v[4], "leaf_rows": v[5], "far_probes": v[6], "blocks": v[7]}, v

    def distinct_nodes(frames_sample, forest_arr, cores):
        """How much of every level the sample's walks really touch (oracle/rdf_oracle.c's visit map; summed over the trees)."""
        from oracle import rdf_oracle
        per = rdf_oracle.distinct_nodes_per_level(frames_sample, forest_arr, n_threads=cores)
        n_trees = per.shape[0]
        tot = per.sum(axis=0)
        return {"frames": int(frames_sample.shape[0]), "per_level": [int(v) for v in tot],
                "share_of_level": [round(float(v) / (n_trees << j), 4) for j, v in enumerate(tot)],
                "hot_record_bytes_touched_below_level_6": int(tot[7:].sum()) * 16 if tot.shape[0] > 7 else 0}

    def leg_cfg5(steps, warmup, check, topology="full"):
        F5, H5, W5, T5, D5, C5 = a.cfg5_frames, 720, 1280, a.cfg5_trees, a.cfg5_depth, 4
        f_np = cached(f"forest_T{T5}_D{D5}_C{C5}_{topology}", lambda: synth.forest(T5, D5, C5, topology))
        fr_np = cached(f"frames_dense_{F5}_{H5}x{W5}_at5000", lambda: synth.frames(["dense"] * F5, 5000, H5, W5))
        forest5 = rdf.DecisionForest.from_numpy(np.asarray(f_np))
        depth5 = rdf.to_device(np.asarray(fr_np))
        lab5 = rdf.DeviceArray((F5, H5, W5), np.uint16).fill(65535)
        tune5 = None
        if not a.unpacked:
            forest5.packed(1.0)
            tune5 = tune_forest(forest5, depth5[0:min(16, F5)], f"T{T5}_D{D5}_C{C5}_{topology}_{H5}x{W5}")
        for _ in range(warmup):
            ev.get_labels_forest(forest5, depth5, lab5)
        e5 = Events(rt, 2 * steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            e5.record(2 * i)
            ev.get_labels_forest(forest5, depth5, lab5)
            e5.record(2 * i + 1)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        kms = float(np.mean([e5.elapsed_ms(2 * i, 2 * i + 1) for i in range(steps)]))
        e5.destroy()
        useful5, _ = useful_lines(forest5, depth5) if check else (None, None)
        res = {"value": round(F5 * H5 * W5 / wall / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(wall * 1e3, 4),
               "kernel_ms": round(kms, 4), "steps": steps, "warmup": warmup, "useful_lines": useful5,
               "workload": f"{F5} dense {W5}x{H5} frames, T{T5}/D{D5}/C{C5} {topology} forest (config 5's per-GPU shard), 1 GPU",
               "tune": tune5}
        if check:
            from oracle import rdf_oracle
            got = lab5[0:2].get()
            want = np.full((2, H5, W5), 65535, np.uint16)
            st = np.zeros(3, np.uint64)
            rdf_oracle.eval_forest(np.asarray(fr_np[0:2]), np.asarray(f_np), want, n_threads=min(host_cores(), rdf_oracle.max_threads()), stats=st)
            mism = int((want != got).sum())
            res["parity"] = {"frames_checked": 2, "pixels": int(want.size), "differing_pixels": mism,
                             "checker": "oracle/rdf_oracle.c on the host"}
            assert mism == 0, f"config 5: GPU labels differ from the oracle in {mism} pixels"
            # algorithmic bytes of the launch: every frame is dense and the topology full, so every pixel
            # visits T*D records and T leaves -- the two checked frames' counters confirm it
            assert int(st[0]) == 2 * H5 * W5 and int(st[1]) == 2 * H5 * W5 * T5 * D5 and int(st[2]) == 2 * H5 * W5 * T5
            res["algorithmic_bytes"] = synth.algorithmic_bytes(F5, H5, W5, 1, False, C5,
                                                               [F5 * H5 * W5, F5 * H5 * W5 * T5 * D5, F5 * H5 * W5 * T5])
            res["distinct_nodes"] = distinct_nodes(np.asarray(fr_np[0:2]), np.asarray(f_np), min(host_cores(), rdf_oracle.max_threads()))
        del forest5, depth5, lab5
        torch.cuda.empty_cache()
        return res

    if a.leg in ("cfg5", "cfg5_balanced"):
        print(json.dumps({"leg": a.leg, **leg_cfg5(a.steps, a.warmup, False, "balanced" if a.leg == "cfg5_balanced" else "full")}), flush=True)
        return

    def leg_cfg5_all_ranks(steps, warmup):
        """Config 5 itself at N > 1 (BASELINE configs[4]: 256 frames on 8 GPUs = this leg at N = 8): every rank evaluates
        its own shard of dense 1280x720 frames with the replicated T8/D22 forest; timed once without and once with the
        label maps gathered to rank 0 (plain RCCL gather, 59 MB per rank and step) inside the timed region; barrier +
        synchronize on both sides, MAX over ranks.  Every rank reaches every collective: a rank that cannot get its inputs
        says so through the first all_reduce and the leg is skipped everywhere."""
        F5, H5, W5, T5, D5, C5 = a.cfg5_frames, 720, 1280, a.cfg5_trees, a.cfg5_depth, 4
        ok, f_np, fr_np, err = 1, None, None, None
        name = f"forest_T{T5}_D{D5}_C{C5}_full"
        try:
            if rank == 0:
                f_np = cached(name, lambda: synth.forest(T5, D5, C5, "full"))   # the others find it in the cache
        except Exception as e:   # noqa: BLE001
            ok, err = 0, repr(e)
        dist.barrier()
        try:
            if rank != 0:
                f_np = cached(name, lambda: synth.forest(T5, D5, C5, "full"))
            fr_np = cached(f"frames_dense_{F5}_{H5}x{W5}_at{5000 + rank * F5}",
                           lambda: synth.frames(["dense"] * F5, 5000 + rank * F5, H5, W5))
        except Exception as e:   # noqa: BLE001
            ok, err = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            return {"skipped": err or "another rank could not build its inputs"}
        forest5 = rdf.DecisionForest.from_numpy(np.asarray(f_np))
        depth5 = rdf.to_device(np.asarray(fr_np))
        lab5 = rdf.DeviceArray((F5, H5, W5), np.uint16).fill(65535)
        tune5 = None
        if not a.unpacked:
            forest5.packed(1.0)
            tune5 = tune_forest(forest5, depth5[0:min(16, F5)])      # (per rank, outside the timed regions; no collective inside)
        mine = lab5.torch_bytes()           # uint8: every backend gathers bytes
        slabs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None

        def run(with_gather):
            for _ in range(warmup):
                ev.get_labels_forest(forest5, depth5, lab5)
                if with_gather:
                    dist.gather(mine, slabs, dst=0)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                ev.get_labels_forest(forest5, depth5, lab5)
                if with_gather:
                    dist.gather(mine, slabs, dst=0)
            sync_all()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        t_kernel = run(False)
        t_gather = run(True)
        # did rank 0 receive every rank's label maps?  (checksums, as for the headline)
        m64 = mine.view(torch.int16).to(torch.int64)
        sums = torch.stack([m64.sum(), (m64 * (torch.arange(m64.numel(), device=m64.device) % 8191)).sum()])
        allsums = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
        check = None
        if rank == 0:
            check = "ok"
            for g in range(world):
                p64 = slabs[g].view(torch.int16).to(torch.int64)
                chk = torch.stack([p64.sum(), (p64 * (torch.arange(p64.numel(), device=p64.device) % 8191)).sum()])
                if not torch.equal(chk, allsums[g]):
                    check = "MISMATCH"
        pix = world * F5 * H5 * W5
        # the same shard with every rank's kernel writing its labels straight into rank 0's IPC-mapped ring (no transfer stage:
        # DESIGN.md section 6, third way).  Collective: a rank on which something raises stops doing work but still walks
        # through every collective below (as `timed` does), the outcome is agreed by an all-gather, and the ring is closed
        # on every path -- peers unmap, barrier, rank 0 frees.
        def direct_stores_leg():
            pg5 = dmod.PeerCopyGather(world, rank, F5 * H5 * W5 * 2)      # (every rank leaves the constructor with the same `ok`)
            if not pg5.ok:
                return {"unavailable": "; ".join(f"rank {g}: {why}" for g, why in (pg5.errors or {}).items())}
            failure = []

            def guarded(fn):
                if failure:
                    return None
                try:
                    return fn()
                except Exception as e:   # noqa: BLE001
                    failure.append(f"rank {rank}: {type(e).__name__}: {e}"[:300])
                    return None
            try:
                pe5 = guarded(lambda: dmod.PeerCopyForestEvaluator(ev, forest5, F5, (H5, W5), pg5, direct_stores=True))
                for _ in range(warmup):
                    guarded(lambda: pe5.step(depth5, None))
                guarded(lambda: pe5.drain())
                sync_all()
                t0 = time.perf_counter()
                for _ in range(steps):
                    guarded(lambda: pe5.step(depth5, None))
                guarded(lambda: pe5.drain())
                sync_all()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                t_direct = float(t.item())
                chk = None
                if rank == 0:
                    def verify():
                        got5 = pe5.result()
                        for g in range(world):
                            p64 = got5[g * F5:(g + 1) * F5].reshape(-1).to(torch.int64)
                            c2_ = torch.stack([p64.sum(), (p64 * (torch.arange(p64.numel(), device=p64.device) % 8191)).sum()])
                            if not torch.equal(c2_, allsums[g]):
                                return "MISMATCH"
                        return "ok"
                    chk = guarded(verify)
                why = [None] * world
                dist.all_gather_object(why, failure[0] if failure else None)
                if any(why):
                    return {"error": "; ".join(w for w in why if w)}
                return {"value": round(pix * steps / t_direct / 1e6, 2), "ms_per_step": round(t_direct / steps * 1e3, 4), "gather_check": chk}
            finally:
                pe5 = None
                dist.barrier()
                if rank != 0:
                    pg5.close()
                dist.barrier()
                if rank == 0:
                    pg5.close()

        try:
            direct = direct_stores_leg()
        except Exception as e:   # noqa: BLE001 -- (a failing collective itself: nothing more can be agreed on)
            direct = {"error": f"rank {rank}: {type(e).__name__}: {e}"[:300]}
        res = {"value": round(pix * steps / t_gather / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(t_gather / steps * 1e3, 4),
               "value_kernel_only": round(pix * steps / t_kernel / 1e6, 2), "kernel_only_ms": round(t_kernel / steps * 1e3, 4),
               "n_gpus": world, "steps": steps, "warmup": warmup, "gather": "rccl gather to rank 0 inside the timed region",
               "gather_check": check, "scaling": "weak", "p2p_direct_stores": direct, "deep_level_table_rank0": tune5,
               "workload": f"{world} x {F5} dense {W5}x{H5} frames, T{T5}/D{D5}/C{C5} full forest replicated "
                           f"(config 5{' itself' if world * F5 == 256 else ' shards'}), {world} GPUs"}
        del forest5, depth5, lab5, slabs
        torch.cuda.empty_cache()
        return res

    # ================================================================================================================
    # headline workload
    # ================================================================================================================
    H, W, F, T, D, C = a.height, a.width, a.frames, a.trees, a.depth, a.classes
    if a.leg == "headline_balanced":        # (the batch of the headline on the balanced forest: what the --pmc passes of cfg2_balanced profile)
        a.topology, a.leg = "balanced", "headline"
    if a.leg == "headline_trainer":         # (... on the forest this run's cfg2_trainer_forest leg trained: round 6)
        a.topology, a.leg = "trainer", "headline"

    def make_forest():
        if a.topology == "trainer":
            raise SystemExit(f"--topology trainer: no forest_T{T}_D{D}_C{C}_trainer in the bench cache (the cfg2_trainer_forest leg of a default run leaves it there)")
        return synth.forest(T, D, C, a.topology)
    forest_np = np.asarray(cached(f"forest_T{T}_D{D}_C{C}_{a.topology}", make_forest))
    forest = rdf.DecisionForest.from_numpy(forest_np)
    frames_np = np.asarray(cached(f"frames_mixed_{F}_{H}x{W}_at{rank * F}", lambda: synth.mixed_batch(F, first_idx=rank * F, h=H, w=W)))
    depth = rdf.to_device(frames_np)
    labels = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
    tune = None
    if not a.unpacked:
        forest.packed(1.0)  # load-time repack, outside the timed region (like the reference's upload)
        tune = tune_forest(forest, depth[0:min(32, F)], f"T{T}_D{D}_C{C}_{a.topology}_{H}x{W}_F{F}")

    # ---- config 2 proper: ONE 848x480 frame per launch (dense frame 0) ----
    def leg_cfg2(n1, forest_obj=None):
        forest_obj = forest if forest_obj is None else forest_obj
        one_d, one_l = depth[0:1], rdf.DeviceArray((1, H, W), np.uint16).fill(65535)
        for _ in range(10):
            ev.get_labels_forest(forest_obj, one_d, one_l)
        e1 = Events(rt, 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e1.record(0)
        for _ in range(n1):
            ev.get_labels_forest(forest_obj, one_d, one_l)
        e1.record(1)
        torch.cuda.synchronize()
        wall1 = (time.perf_counter() - t1) / n1
        dev1 = e1.elapsed_ms(0, 1) / n1 / 1e3
        e1.destroy()
        return {"value": round(H * W / wall1 / 1e6, 2), "unit": "Mpix/s", "ms_per_frame_wall": round(wall1 * 1e3, 4),
                "ms_per_frame_device": round(dev1 * 1e3, 4), "kernel_ms": round(dev1 * 1e3, 4), "frame": "dense #0",
                "launches": n1}

    if a.leg == "cfg2":
        print(json.dumps({"leg": "cfg2", **leg_cfg2(50)}), flush=True)
        return

    # N>1: one launch per step; the gather of step s overlaps the evaluation of step s+1 (two label buffers);
    # --chunks C > 0 selects the in-step pipeline instead (C launches, gather of chunk c overlaps chunk c+1)
    chunks = a.chunks or 1
    overlapped = multi and a.chunks == 0
    sharded = dmod.ShardedForestEvaluator(ev, forest, F, (H, W), n_chunks=chunks)
    ring = [labels, rdf.DeviceArray((F, H, W), np.uint16).fill(65535)] if overlapped else None

    # N>1, default: peer copies over xGMI by the copy engines (RCCL carries only the control plane); needs HIP IPC
    # between the ranks' processes, falls back to the RCCL gather if any rank cannot map rank 0's buffer
    peer, pg, peer_direct, pg_d, notes = None, None, None, None, {}      # notes: what was unavailable on this run and why (goes on the N > 1 line)
    if multi and a.chunks == 0 and a.gather in ("auto", "fastest", "p2p", "both"):
        pg = dmod.PeerCopyGather(world, rank, F * H * W * 2, _fail_open_on_rank=a.fail_ipc_open_on_rank)
        if pg.ok:
            peer = dmod.PeerCopyForestEvaluator(ev, forest, F, (H, W), pg)
        else:
            notes["p2p"] = "unavailable: " + "; ".join(f"rank {g}: {why}" for g, why in (pg.errors or {}).items())
        # the same ring a second time for the mode in which every rank's kernel writes its labels straight into rank 0's
        # memory (no copy at all): its own ring, so that the two modes' step counters do not meet
        if pg.ok and a.gather in ("auto", "fastest", "both"):
            pg_d = dmod.PeerCopyGather(world, rank, F * H * W * 2)
            if pg_d.ok:
                peer_direct = dmod.PeerCopyForestEvaluator(ev, forest, F, (H, W), pg_d, direct_stores=True)
            else:
                notes["p2p direct stores"] = "unavailable: " + "; ".join(f"rank {g}: {why}" for g, why in (pg_d.errors or {}).items())

    # ---- algorithmic bytes of one step, from the visit counters of the same walk ----
    dstats = rdf.DeviceArray((3,), np.uint64).fill(0)
    scratch = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
    rc = lib.rdf_eval_forest_stats(depth.ptr, F, W, H, forest.forest_cu.ptr, T, D, C, None, -1, scratch.ptr, 1, 1.0,
                                   dstats.ptr, rt.stream())
    assert rc == 0, lib.rdf_error_string(rc)
    stats = dstats.get()
    alg_bytes = synth.algorithmic_bytes(F, H, W, 1, False, C, stats)

    # ---- a compute stream that leaves one CU per shader engine to RCCL (its send/recv kernel cannot start beside the
    # forest kernel's persistent workgroups otherwise: tools/ubench_overlap.py) ----
    def masked_stream(reserve):
        import ctypes
        h = ctypes.c_void_p()
        try:
            rc_m = lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), reserve)
            why = None if rc_m == 0 else str(lib.rdf_error_string(rc_m))
            st_m = torch.cuda.ExternalStream(h.value) if rc_m == 0 else None
        except Exception as e:   # noqa: BLE001
            why, st_m = repr(e), None
        if st_m is None:
            print(f"rank {rank}: no CU-masked stream ({why}); using the current stream", file=sys.stderr)
            return None, None, why
        return h, st_m, None

    def timed(step_fn_raw, drain_fn_raw, stream):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks.  A step that
        raises on this rank stops doing work but the rank keeps walking through every barrier, so that the other ranks are
        not left waiting; the mode is then reported as failed (third return value) on every rank."""
        failure = []

        def guarded(fn):
            def run():
                if failure:
                    return
                try:
                    fn()
                except Exception as e:   # noqa: BLE001
                    failure.append(f"rank {rank}: {type(e).__name__}: {e}"[:300])
            return run
        step_fn, drain_fn = guarded(step_fn_raw), guarded(drain_fn_raw)
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        with ctx:
            for _ in range(a.warmup):
                step_fn()
            drain_fn()
            evs = Events(rt, 2 * a.steps)
            sync_all()
            t0 = time.perf_counter()
            for i in range(a.steps):
                evs.record(2 * i)
                step_fn()
                evs.record(2 * i + 1)
            drain_fn()              # every step's label maps are on rank 0 before the clock stops
            sync_all()
            elapsed = time.perf_counter() - t0
            kms = [evs.elapsed_ms(2 * i, 2 * i + 1) for i in range(a.steps)]
            evs.destroy()
        failed = None
        if multi:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            why = [None] * world
            dist.all_gather_object(why, failure[0] if failure else None)
            if any(why):
                failed = "; ".join(w for w in why if w)
        elif failure:
            raise RuntimeError(failure[0])
        return elapsed, kms, failed

    def gather_check(result_fn, from_peer):
        """Did rank 0 really receive every rank's label maps?  (checksum of checksums; every rank's reference is `scratch`,
        the labels the visit-counter kernel wrote for the same frames: another kernel, no transfer)"""
        mine = scratch.torch_bytes().view(torch.int16).to(torch.int64)
        sums = torch.stack([mine.sum(), (mine * (torch.arange(mine.numel(), device=mine.device) % 8191)).sum()])
        allsums = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
        if rank != 0:
            return None
        got = result_fn()
        ok = True
        for g in range(world):
            part = got[g * F:(g + 1) * F]
            part = (part.reshape(-1) if from_peer else part.torch_bytes().view(torch.int16)).to(torch.int64)
            chk = torch.stack([part.sum(), (part * (torch.arange(part.numel(), device=part.device) % 8191)).sum()])
            ok = ok and bool(torch.equal(chk, allsums[g]))
        return "ok" if ok else "MISMATCH"

    modes = {}          # name -> (step, drain, stream, reserve, result_fn, from_peer)
    handles = []

    def rccl_mode():
        reserve = a.reserve_cus if a.reserve_cus >= 0 else 32
        h, st, why = masked_stream(reserve) if reserve > 0 else (None, None, None)
        if why:
            notes["cu_masked_stream"] = f"unavailable: {why}"
        handles.append(h)
        # split steps (round 6): the CUs the masked stream leaves to RCCL are idle once the previous step's gather has finished;
        # a helper launch that waits for that gather takes them back and pulls tiles from the main launch's queue
        sharded.helper_cus = reserve if (st is not None and overlapped and not a.no_split_step and not a.unpacked) else 0
        step = (lambda: sharded.step_overlapped(depth, ring)) if overlapped else (lambda: sharded.step(depth, labels))
        return (step, sharded.drain, st, reserve if st is not None else 0, sharded.result, False)

    def over_budget(what):
        """N>1: has the run used up its wall-time budget?  Rank 0's clock decides for every rank (one broadcast), and what is
        skipped is named on the line (distributed.unavailable)."""
        if not multi:
            return False
        flag = [bool(time.perf_counter() - t_main > a.time_budget)]
        dist.broadcast_object_list(flag, src=0)
        if flag[0]:
            notes[what] = f"skipped: the {a.time_budget:.0f}-s wall-time budget of the N>1 run was spent before it (--time-budget)"
        return flag[0]

    if not multi:
        modes["none"] = (lambda: sharded.step(depth, labels), sharded.drain, None, 0, None, False)
    else:
        # (the RCCL gather first where it is `value` -- every choice but "both" and "p2p": it is never the part the wall-time
        # guard skips)
        want_rccl = peer is None or a.gather in ("rccl", "both", "auto", "fastest")
        if want_rccl and a.gather != "both":
            modes["rccl gather"] = rccl_mode()
        if peer is not None and a.gather != "rccl":
            modes["p2p copy engines"] = (lambda: peer.step(depth, ring), peer.drain, None, 0, peer.result, True)
        if peer_direct is not None:
            modes["p2p direct stores"] = (lambda: peer_direct.step(depth, None), peer_direct.drain, None, 0, peer_direct.result, True)
        if want_rccl and "rccl gather" not in modes:
            modes["rccl gather"] = rccl_mode()
    torch.cuda.synchronize()

    results, failed_modes = {}, {}
    queue = list(modes.items())
    while queue:
        name, (step, drain, st, reserve, result_fn, from_peer) = queue.pop(0)
        if results and over_budget(name):           # (never the first mode)
            continue
        elapsed, kms, failed = timed(step, drain, st)
        if failed is None and multi:
            try:
                check = gather_check(result_fn, from_peer)
            except Exception as e:   # noqa: BLE001 -- (every rank has left the collectives of gather_check by now or none has)
                check = f"check failed: {type(e).__name__}: {e}"[:200]
        else:
            check = None
        if failed is not None:
            failed_modes[name] = failed
            notes[name] = f"failed: {failed}"
            if name == "p2p copy engines" and "rccl gather" not in modes:     # (agreed on every rank: `failed` is all-gathered)
                modes["rccl gather"] = rccl_mode()
                queue.append(("rccl gather", modes["rccl gather"]))
            continue
        results[name] = {"elapsed": elapsed, "kern_ms": kms, "reserve": reserve, "gather_check": check}
        if name == "rccl gather" and sharded.helper_cus > 0:
            results[name]["split_step_helper_workgroups"] = sharded.helper_workgroups
        if name in ("p2p copy engines", "p2p direct stores") and rank == 0:
            # the ring's ready counters after the drain: the last two steps of every rank are marked as landed
            try:
                pe_, pg_ = (peer, pg) if name == "p2p copy engines" else (peer_direct, pg_d)
                last = pe_.steps_done - 1
                cnt = pg_.ready_counters()
                results[name]["ready_counters_ok"] = bool((cnt[last % pg_.n_slots] == last + 1).all() and
                                                          (pg_.n_slots < 2 or last < 1 or (cnt[(last - 1) % pg_.n_slots] == last).all()))
            except Exception as e:   # noqa: BLE001
                results[name]["ready_counters_ok"] = f"not read: {e}"[:120]
    if not results:
        raise SystemExit(f"every gather mode failed: {failed_modes}")
    # `value` is the RCCL gather's by default (north_star: "a single RCCL gather of the label maps over xGMI"); the other
    # transports are timed beside it and reported (distributed.gather_modes).  --gather fastest: the fastest intact mode's;
    # p2p | rccl | both: the first mode's that ran.  A mode whose label maps did not arrive intact is never `value`.
    primary = next(iter(results))
    if a.gather in ("auto", "fastest") and len(results) > 1:
        # (`elapsed` is the MAX over ranks and the check's verdict is broadcast below: the same choice on every rank)
        verdicts = [{n: r["gather_check"] for n, r in results.items()}]
        dist.broadcast_object_list(verdicts, src=0)
        intact = [n for n in results if verdicts[0].get(n) == "ok"] or list(results)
        if a.gather == "auto" and "rccl gather" in intact:
            primary = "rccl gather"
        else:
            primary = min(intact, key=lambda n: results[n]["elapsed"])
    elapsed, kern_ms = results[primary]["elapsed"], results[primary]["kern_ms"]
    if not multi:
        assert np.array_equal(labels.get(), scratch.get()), "timed path and stats path disagree"

    # ---- the kernel alone on every rank (no gather enqueued): Mpix/s with and without the gather, SURVEY 8(e) ----
    kern_only = None
    if multi:
        ek = Events(rt, 2)
        sync_all()
        ek.record(0)
        for _ in range(a.steps):
            ev.get_labels_forest(forest, depth, labels)
        ek.record(1)
        t = torch.tensor([ek.elapsed_ms(0, 1) / a.steps], dtype=torch.float64, device="cuda")
        ek.destroy()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kern_only = float(t.item())
        assert np.array_equal(labels.get(), scratch.get()), "timed path and stats path disagree"

    pix_per_step = world * F * H * W
    value = pix_per_step * a.steps / elapsed / 1e6
    kern_avg_ms = float(np.mean(kern_ms))
    kern_med_ms = float(np.median(kern_ms))

    if a.leg == "headline":
        print(json.dumps({"leg": "headline", "topology": a.topology, "kernel_ms": round(kern_avg_ms, 4), "value": round(value, 2), "tune": tune}), flush=True)
        return

    gather_mode = None if not multi else primary
    # N = 1: `value` is the MEDIAN of the K per-step hipEvent times (SURVEY 8d's protocol), the K steps between the two
    # barriers -- what the bench contract brackets -- are beside it as value_mean / ms_per_step_mean (they differ by a few
    # tenths of a per cent).  N > 1: `value` is the bracketed K steps, MAX over ranks, gather included; a per-step median
    # would leave the gather out, so it is reported for this rank's launches only (ms_per_step_median).
    mean_ms = elapsed / a.steps * 1e3
    if not multi:
        head = {"value": round(F * H * W / kern_med_ms / 1e3, 2), "ms_per_step": round(kern_med_ms, 4),
                "value_mean": round(value, 2), "ms_per_step_mean": round(mean_ms, 4), "value_is": "median of the per-step times"}
    else:
        head = {"value": round(value, 2), "ms_per_step": round(mean_ms, 4), "ms_per_step_median": round(kern_med_ms, 4),
                "value_is": "K steps between barriers, MAX over ranks, gather included"}
    out_value_tmp = head["value"]
    valid_px = int(stats[0])          # pixels the forest really evaluates (the rest is background / holes: tree_eval.cu:88-89)
    out = {
        "metric": "classified Mpix/s on 848x480 depth frames (4 trees, depth 20); % of the bounding roofline level "
                  "(roofline.bound names it; the HBM level is roofline.levels.hbm.frac)",
        "value": head.pop("value"), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": head.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", **head,
        "config": {"workload": f"{F} x {W}x{H} depth frames per GPU per step (config 4 shard = config 2 frame x {F}; "
                               f"half dense, half live-like), T{T}/D{D}/C{C} {a.topology} forest, "
                               + (f"labels gathered to rank 0 ({gather_mode}; control plane {a.backend}) inside the timed region" if multi else "1 GPU"),
                   "frames_per_gpu": F, "frame": [H, W], "trees": T, "tree_depth": D, "classes": C,
                   "topology": a.topology, "forest_layout": "reference" if a.unpacked else "packed16",
                   "pipeline_chunks": chunks, "tile_schedule": a.scheduler, "cus_left_to_rccl": results[primary]["reserve"],
                   "gather": gather_mode,
                   "gather_overlap": ("next step" if (overlapped or peer is not None) else ("in-step chunks" if multi else None)),
                   "sharding": f"frames x{world}, forest replicated", "gather_check": results[primary]["gather_check"],
                   "deep_level_table": tune,
                   # which kernels these numbers are of: the id baked into librdf_hip.so (a hash of its sources and flags)
                   "library_build_id": (lambda b: b.decode() if isinstance(b, bytes) else str(b))(lib.rdf_build_id())},
        # the same rate counted over the pixels the forest really evaluates (this rank's batch: half of it is live-like
        # frames, 85 % background); the metric counts every depth pixel
        "value_valid_pixels": round(out_value_tmp * valid_px / (F * H * W), 2),
        "valid_pixel_share": round(valid_px / (F * H * W), 4),
    }
    if multi:
        # what a driver needs to verify the run: N ranks of ONE RCCL communicator on N distinct devices
        uuid = str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", f"index-{dev_index}"))
        ids = [None] * world
        dist.all_gather_object(ids, {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "uuid": uuid,
                                     "name": torch.cuda.get_device_name(dev_index)})
        out["distributed"] = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "devices": ids,
                              "distinct_devices": len({d["uuid"] for d in ids}),
                              "kernel_only_ms": round(kern_only, 4),
                              "value_kernel_only": round(pix_per_step / (kern_only * 1e-3) / 1e6, 2),
                              "gather_modes": {n: {"ms_per_step": round(r["elapsed"] / a.steps * 1e3, 4),
                                                   "value": round(pix_per_step * a.steps / r["elapsed"] / 1e6, 2),
                                                   "cus_left_to_rccl": r["reserve"], "gather_check": r["gather_check"],
                                                   **({"split_step_helper_workgroups": r["split_step_helper_workgroups"]}
                                                      if "split_step_helper_workgroups" in r else {}),
                                                   **({"ready_counters_ok": r["ready_counters_ok"]} if "ready_counters_ok" in r else {})}
                                               for n, r in results.items()},
                              # what this run could not use, and why (p2p: the receive ring could not be mapped by every rank;
                              # cu_masked_stream: the RCCL gather then ran beside a kernel on all CUs; a mode that failed mid-run)
                              "unavailable": notes}

        if not a.no_cfg5 and not over_budget("cfg5_all_ranks"):
            try:
                c5n = leg_cfg5_all_ranks(5, 2)       # every rank takes part
            except Exception as e:   # noqa: BLE001 -- the headline is measured; it must still be printed
                c5n = {"error": f"rank {rank}: {type(e).__name__}: {e}"[:400]}
            out["cfg5_all_ranks"] = c5n

    key = f"F{F}_T{T}_D{D}_C{C}_{a.topology}"
    useful_h, st8_h = useful_lines(forest, depth) if full_run else (None, None)
    if st8_h is not None:       # the packed launch's own counters agree with the reference-layout stats kernel's
        assert st8_h[0:3] == [int(stats[0]), int(stats[1]), int(stats[2])], (st8_h, [int(v) for v in stats])
    out["roofline"] = roofline_for("headline", key, live, kern_avg_ms, alg_bytes, useful_h)
    if st8_h is not None:
        out["roofline"]["node_records_from_lds"] = st8_h[3]
    out["roofline"]["algorithmic"].update({"bytes_per_pixel": round(alg_bytes / (F * H * W), 1),
                                           "visits": {"pixels": int(stats[0]), "node_records": int(stats[1]), "leaves": int(stats[2])}})

    # (top level: what bounds the headline kernel and how far the launch is from the HBM roofline)
    out["bound"] = out["roofline"].get("bound")
    out["hbm_frac"] = (out["roofline"].get("levels", {}).get("hbm") or {}).get("frac")

    if rank == 0 and full_run:
        # ---- what HBM really delivers to a plain stream: a 1-GB device-to-device copy ----
        src_t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").fill_(1)
        dst_t = torch.empty_like(src_t)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(4):
            c0.record()
            dst_t.copy_(src_t)
            c1.record()
            torch.cuda.synchronize()
            ms = c0.elapsed_time(c1)
            best = ms if best is None else min(best, ms)
        ceiling = 2.0 * (1 << 30) / (best * 1e-3) / 1e9      # bytes read + bytes written
        out["roofline"]["copy_ceiling"] = {"value": round(ceiling, 1), "unit": "GB/s",
                                           "what": "1-GB device-to-device copy, read + written bytes per second"}
        del src_t, dst_t

        c2 = leg_cfg2(200)
        one_stats = rdf.DeviceArray((3,), np.uint64).fill(0)
        one_l = rdf.DeviceArray((1, H, W), np.uint16).fill(65535)
        assert lib.rdf_eval_forest_stats(depth.ptr, 1, W, H, forest.forest_cu.ptr, T, D, C, None, -1, one_l.ptr, 1, 1.0,
                                         one_stats.ptr, rt.stream()) == 0
        c2["roofline"] = roofline_for("cfg2", f"F1_T{T}_D{D}_C{C}_{a.topology}", live, c2["kernel_ms"],
                                      synth.algorithmic_bytes(1, H, W, 1, False, C, one_stats.get()))
        out["cfg2_single_frame"] = c2

        # ---- config 3: 2-layer stack through LayeredDecisionForest.run (run_live_layered.py:126), r = 2 ----
        cfg3 = {"layers": [{"model": forest}, {"model": forest, "filter_model": 0, "filter_model_class": 3}],
                "conditions": [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5], [0, 6]],
                "label_colors": [[0, 0, 0, 255]] * 6}
        lf = rdf.LayeredDecisionForest(cfg3, (H, W), 2)
        dbuf, lbuf = rdf.GpuBuffer((H, W), np.uint16), rdf.GpuBuffer((H // 2, W // 2), np.uint16)
        dbuf.cu().set(frames_np[1 if F > 1 else 0])   # a live-like frame
        for _ in range(10):
            lf.run(dbuf, lbuf, 1.0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(100):
            lf.run(dbuf, lbuf, 1.0)
        torch.cuda.synchronize()
        w3 = (time.perf_counter() - t3) / 100
        # what the calls cost on the HOST (rdf_debug_host_overhead): a sync-per-frame live loop pays this per frame
        import ctypes
        ho = (ctypes.c_ulonglong * 5)()
        lib.rdf_debug_host_overhead(ho, 1)
        for _ in range(200):
            lf.run(dbuf, lbuf, 1.0)
        lib.rdf_debug_host_overhead(ho, 1)
        one_l3 = rdf.DeviceArray((1, H, W), np.uint16).fill(65535)
        ho2 = (ctypes.c_ulonglong * 5)()
        for _ in range(200):
            ev.get_labels_forest(forest, depth[0:1], one_l3)
        lib.rdf_debug_host_overhead(ho2, 1)
        torch.cuda.synchronize()
        host_overhead = {"rdf_layered_run_us_per_call_runtime_included": round(ho[3] / max(1, ho[4]) / 1e3, 2), "rdf_layered_run_calls": int(ho[4]),
                         "rdf_eval_forest_packed_us_in_library_before_the_launch": round(ho2[0] / max(1, ho2[2]) / 1e3, 2),
                         "rdf_eval_forest_packed_us_in_hip_launch": round(ho2[1] / max(1, ho2[2]) / 1e3, 2), "rdf_eval_forest_packed_calls": int(ho2[2]),
                         "what": "CLOCK_MONOTONIC inside librdf_hip.so (rdf_debug_host_overhead), 200 calls each"}
        out["cfg3_layered_run"] = {"ms_per_frame_wall": round(w3 * 1e3, 4), "value": round(H * W / w3 / 1e6, 2), "host_overhead": host_overhead,
                                   "unit": "Mpix/s", "what": "LayeredDecisionForest.run, 2 layers (second filtered on "
                                   "class 3 of the first), labels_reduce 2, one live-like 848x480 frame: ONE C-ABI call, both "
                                   "layers in one forest launch (unfiltered; the composite kernel applies the filter and "
                                   "the reference's three fills are folded in)"}

        # the live loop of run_live_layered.py:66-126 with the frame arriving in (pinned) host memory and the composite wanted
        # there: upload, run, and the result either read back (the reference's `.get()`) or written to host memory by the
        # composite kernel itself (GpuBuffer(host_mapped=True)); every frame is waited for, as a live loop does
        if not a.no_pcie:
            one_pin = torch.from_numpy(np.array(frames_np[1 if F > 1 else 0]).view(np.int16).reshape(-1)).pin_memory()
            d1_t = dbuf.cu().torch_bytes().view(torch.int16)
            hbuf = rdf.GpuBuffer((H // 2, W // 2), np.uint16, host_mapped=True)
            back = torch.empty((H // 2) * (W // 2), dtype=torch.int16).pin_memory()
            l1_t = lbuf.cu().torch_bytes().view(torch.int16)
            live_ms = {}
            for name in ("read_back", "written_by_the_kernel"):
                for rep in range(110):
                    if rep == 10:
                        torch.cuda.synchronize()
                        tl = time.perf_counter()
                    d1_t.copy_(one_pin, non_blocking=True)
                    if name == "read_back":
                        lf.run(dbuf, lbuf, 1.0)
                        back.copy_(l1_t, non_blocking=True)
                    else:
                        lf.run(dbuf, hbuf, 1.0)
                    torch.cuda.synchronize()
                live_ms[name] = (time.perf_counter() - tl) / 100
            same3 = bool(np.array_equal(hbuf.host.reshape(-1), back.numpy().view(np.uint16)))
            out["cfg3_layered_run"].update({
                "ms_per_frame_wall_from_host_read_back": round(live_ms["read_back"] * 1e3, 4),
                "ms_per_frame_wall_from_host_composite_written_to_host": round(live_ms["written_by_the_kernel"] * 1e3, 4),
                "from_host_composites_equal": same3,
                "from_host_what": "per frame: pinned H2D of the frame, LayeredDecisionForest.run, composite on the host (copied "
                                  "back / written there by the composite kernel), host waits for every frame"})

        # ---- host-buffer variant: pinned H2D of the frames + kernel + D2H of the labels (never `value`) ----
        if not a.no_pcie:
            pin_in = torch.from_numpy(np.array(frames_np).view(np.int16).reshape(-1)).pin_memory()   # (a writable copy: the cache is memory-mapped read-only)
            pin_out = torch.empty(F * H * W, dtype=torch.int16).pin_memory()
            d_t, l_t = depth.torch_bytes().view(torch.int16), labels.torch_bytes().view(torch.int16)
            for rep in range(4):
                if rep == 1:
                    torch.cuda.synchronize()
                    tp = time.perf_counter()
                d_t.copy_(pin_in, non_blocking=True)
                ev.get_labels_forest(forest, depth, labels)
                pin_out.copy_(l_t, non_blocking=True)
            torch.cuda.synchronize()
            wall_p = (time.perf_counter() - tp) / 3
            out["pcie_inclusive"] = {"value": round(F * H * W / wall_p / 1e6, 2), "unit": "Mpix/s",
                                     "ms_per_step": round(wall_p * 1e3, 3),
                                     "what": "pinned H2D of the batch + kernel + D2H of the labels, serial on one stream"}
            # the same work through HostFramesEvaluator (3d-beats_amd/host_stream.py): the next batch goes up on a copy engine
            # while this one is evaluated, and the kernel writes its labels straight into pinned host memory -- no download
            # stage, frames up and labels down share the link at the same time.  (Round 3's first version -- four chunks
            # through ONE device buffer, labels downloaded by a copy engine -- took 7.5 ms: every upload waited for the kernel
            # reading its slot, and H2D next to D2H take the sum of both, tools/pcie_overlap_probe.py.)
            hp = rdf.HostFramesEvaluator(forest, (F, H, W), evaluator=ev)
            hp.mark_steps = True
            for b_ in range(2):
                hp.frames[b_][:] = frames_np
            reps = 12
            last = None
            for rep in range(reps + 2):
                if rep == 2:
                    torch.cuda.synchronize()
                    tp = time.perf_counter()
                hp.next_frames()             # (the slot already holds the batch; a caller would write the next frames here)
                last = hp.submit()
            pipelined_labels = hp.result(last)
            torch.cuda.synchronize()
            step_marks = hp.step_marks
            wall_pp = (time.perf_counter() - tp) / reps
            # steady state: the intervals between the starts of consecutive steps on the evaluate stream, without the first
            # measured step (it starts on an empty pipeline) -- the wall figure over all of them, fill and drain included, is beside it
            gaps = [step_marks[i].elapsed_time(step_marks[i + 1]) for i in range(3, len(step_marks) - 1)]
            step_ms, step_mean = float(np.median(gaps)), float(np.mean(gaps))
            same = bool(np.array_equal(pipelined_labels, scratch.get()))
            out["pcie_inclusive_pipelined"] = {"value": round(F * H * W / step_ms / 1e3, 2), "unit": "Mpix/s",
                                               "ms_per_step": round(step_ms, 3), "labels_match": same,
                                               "value_is": "median interval between consecutive steps in steady state",
                                               "value_mean": round(F * H * W / step_mean / 1e3, 2),
                                               "ms_per_step_mean": round(step_mean, 3),
                                               "ms_per_step_wall_with_fill_and_drain": round(wall_pp * 1e3, 3),
                                               "steps": reps, "step_intervals_ms": [round(g, 2) for g in gaps],
                                               "what": "HostFramesEvaluator: frames from pinned host memory, uploaded by a copy engine "
                                                       "while the batch before is evaluated; labels written into pinned host memory by "
                                                       "the kernel itself"}

        if not a.no_cpu_baseline:
            from oracle import rdf_oracle  # the checker; never the thing measured as `value`
            got = labels.get()
            cores = min(host_cores(), rdf_oracle.max_threads())
            # (i) parity: every frame of the timed step against the oracle (bounded by --cpu-seconds)
            done, t_par, mism = 0, 0.0, 0
            for i in range(F):      # the batch alternates dense / live-like frames
                want = np.full((1, H, W), 65535, np.uint16)
                tc = time.perf_counter()
                rdf_oracle.eval_forest(frames_np[i:i + 1], forest_np, want, n_threads=cores)
                t_par += time.perf_counter() - tc
                mism += int((want[0] != got[i]).sum())
                done += 1
                if t_par >= a.cpu_seconds:
                    break
            # (ii) the baseline figure: median of 5 passes over a FIXED sample, the step's first 8 frames (4 dense, 4 live-like)
            ns = min(8, F)
            want8 = np.full((ns, H, W), 65535, np.uint16)
            passes = []
            for _ in range(5):
                want8[:] = 65535
                tc = time.perf_counter()
                rdf_oracle.eval_forest(frames_np[0:ns], forest_np, want8, n_threads=cores)
                passes.append(time.perf_counter() - tc)
            mism += int((want8 != got[0:ns]).sum())
            t_med = float(np.median(passes))
            out["cpu_baseline"] = {"value": round(ns * H * W / t_med / 1e6, 3), "unit": "Mpix/s", "cores": cores,
                                   "kind": "port",
                                   "sample": f"median of 5 passes over the step's first {ns} frames (alternating dense/live-like), "
                                             f"{t_med:.2f} s per pass of oracle/rdf_oracle.c (OpenMP, -O2) on the host; parity: "
                                             f"{done} of the step's {F} frames checked in {t_par:.1f} s, GPU labels differ in {mism} pixels",
                                   "passes_s": [round(x, 3) for x in passes], "parity_frames": done, "differing_pixels": mism}
            assert mism == 0, f"GPU labels differ from the oracle in {mism} pixels"
            # what a "numpy fallback" would have been (variant ii): level-synchronous numpy, one process
            from oracle import rdf_numpy
            wn = np.full((1, H, W), 65535, np.uint16)
            tn = time.perf_counter()
            rdf_numpy.eval_forest(frames_np[1:2], forest_np, wn)
            tn = time.perf_counter() - tn
            out["cpu_baseline_numpy"] = {"value": round(H * W / tn / 1e6, 3), "unit": "Mpix/s", "cores": 1,
                                         "kind": "port", "sample": f"1 live-like frame, {tn:.1f} s of oracle/rdf_numpy.py; "
                                         f"labels differ from the GPU's in {int((wn[0] != got[1]).sum())} pixels"}

        # ---- the legs beside the packed forest kernel, each with its own in-run check; a leg that fails becomes a field ----
        def leg(name, fn):
            t_leg = time.perf_counter()
            try:
                out[name] = fn()
            except Exception as e:   # noqa: BLE001 -- the headline must still be printed
                out[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
            out[name]["leg_seconds"] = round(time.perf_counter() - t_leg, 1)

        def leg_unpacked():
            """The same batch straight from the reference-layout forest (the .npy as it is, no rdf_forest_pack)."""
            ev_u = rdf.DecisionTreeEvaluator(use_packed=False)
            lab_u = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
            for _ in range(2):
                ev_u.get_labels_forest(forest, depth, lab_u)
            eu = Events(rt, 10)
            for i in range(5):
                eu.record(2 * i)
                ev_u.get_labels_forest(forest, depth, lab_u)
                eu.record(2 * i + 1)
            ms_u = float(np.median([eu.elapsed_ms(2 * i, 2 * i + 1) for i in range(5)]))
            eu.destroy()
            same = bool(np.array_equal(lab_u.get(), scratch.get()))
            assert same, "reference-layout path and packed path disagree"
            return {"value": round(F * H * W / ms_u / 1e3, 2), "unit": "Mpix/s", "ms_per_step": round(ms_u, 4), "steps": 5,
                    "labels_equal_packed_path": same, "what": "rdf_eval_forest on the reference's float32 [T][2^D-1][7+2C] records"}

        def leg_trained():
            """Config 2 with the trained-like topology (15 % of the sides become leaves per level from level 3 on; PDFs =
            normalised counts): one dense frame per launch, the batch rate, and frames checked against the oracle."""
            from oracle import rdf_oracle
            ft_np = np.asarray(cached(f"forest_T{T}_D{D}_C{C}_trained", lambda: synth.forest(T, D, C, "trained")))
            ft = rdf.DecisionForest.from_numpy(ft_np)
            ft.packed(1.0)
            tune_t = tune_forest(ft, depth[0:min(32, F)])
            res = leg_cfg2(200, ft)
            lab_t = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
            for _ in range(2):
                ev.get_labels_forest(ft, depth, lab_t)
            et = Events(rt, 20)
            for i in range(10):
                et.record(2 * i)
                ev.get_labels_forest(ft, depth, lab_t)
                et.record(2 * i + 1)
            ms_t = float(np.median([et.elapsed_ms(2 * i, 2 * i + 1) for i in range(10)]))
            et.destroy()
            ns = min(8, F)
            want = np.full((ns, H, W), 65535, np.uint16)
            st = np.zeros(3, np.uint64)
            rdf_oracle.eval_forest(frames_np[0:ns], ft_np, want, n_threads=min(host_cores(), rdf_oracle.max_threads()), stats=st)
            mism = int((want != lab_t[0:ns].get()).sum())
            assert mism == 0, f"trained-like topology: GPU labels differ from the oracle in {mism} pixels"
            res.update({"topology": "trained", "tune": tune_t, "batch": {"value": round(F * H * W / ms_t / 1e3, 2), "unit": "Mpix/s",
                                                          "ms_per_step": round(ms_t, 4), "frames": F, "steps": 10},
                        "mean_levels_per_pixel_and_tree": round(float(st[1]) / max(1.0, float(st[0]) * T), 2),
                        "parity": {"frames_checked": ns, "differing_pixels": mism, "checker": "oracle/rdf_oracle.c on the host"}})
            return res

        def leg_balanced():
            """Config 2 / config 4's shard on a forest whose deep levels are OCCUPIED (synth's balanced topology: every
            threshold is the median of its feature over calibration frames, as a trained split is data-adapted; the
            headline's "full" topology draws thresholds that send most pixels one way and touches 1.7 % of level 19): one
            dense frame per launch, the batch rate, frames checked against the oracle, how much of each level the
            batch really visits, and the roofline from this run's own counter passes."""
            from oracle import rdf_oracle
            fb_np = np.asarray(cached(f"forest_T{T}_D{D}_C{C}_balanced", lambda: synth.forest(T, D, C, "balanced")))
            fb = rdf.DecisionForest.from_numpy(fb_np)
            fb.packed(1.0)
            tune_b = tune_forest(fb, depth[0:min(32, F)], f"T{T}_D{D}_C{C}_balanced_{H}x{W}_F{F}")
            res = leg_cfg2(200, fb)
            lab_b = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
            for _ in range(2):
                ev.get_labels_forest(fb, depth, lab_b)
            eb = Events(rt, 20)
            for i in range(10):
                eb.record(2 * i)
                ev.get_labels_forest(fb, depth, lab_b)
                eb.record(2 * i + 1)
            ms_b = float(np.median([eb.elapsed_ms(2 * i, 2 * i + 1) for i in range(10)]))
            eb.destroy()
            ns = min(8, F)
            cores = min(host_cores(), rdf_oracle.max_threads())
            want = np.full((ns, H, W), 65535, np.uint16)
            st = np.zeros(3, np.uint64)
            tc = time.perf_counter()
            rdf_oracle.eval_forest(frames_np[0:ns], fb_np, want, n_threads=cores, stats=st)
            t_cpu = time.perf_counter() - tc
            mism = int((want != lab_b[0:ns].get()).sum())
            assert mism == 0, f"balanced topology: GPU labels differ from the oracle in {mism} pixels"
            # every walk of this topology reaches level D-1: the batch's visit counters are the headline's
            alg_b = synth.algorithmic_bytes(F, H, W, 1, False, C, [int(stats[0]), int(stats[0]) * T * D, int(stats[0]) * T])
            res.update({"topology": "balanced", "tune": tune_b,
                        "batch": {"value": round(F * H * W / ms_b / 1e3, 2), "unit": "Mpix/s", "ms_per_step": round(ms_b, 4),
                                  "value_valid_pixels": round(int(stats[0]) / ms_b / 1e3, 2), "frames": F, "steps": 10},
                        "parity": {"frames_checked": ns, "differing_pixels": mism, "checker": "oracle/rdf_oracle.c on the host"},
                        "cpu_baseline": {"value": round(ns * H * W / t_cpu / 1e6, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
                                         "sample": f"one pass over the batch's first {ns} frames, {t_cpu:.2f} s"},
                        "distinct_nodes": distinct_nodes(frames_np[0:ns], fb_np, cores),
                        "distinct_nodes_full_topology": distinct_nodes(frames_np[0:ns], forest_np, cores) if a.topology == "full" else None})
            useful_b, st8_b = useful_lines(fb, depth)
            if st8_b is not None:
                assert st8_b[0:3] == [int(stats[0]), int(stats[0]) * T * D, int(stats[0]) * T], st8_b
            res["batch"]["roofline"] = roofline_for("headline_balanced", f"F{F}_T{T}_D{D}_C{C}_balanced", live, ms_b, alg_b, useful_b)
            return res

        if not a.no_balanced and not a.unpacked:
            leg("cfg2_balanced", leg_balanced)
            out["value_balanced"] = (out["cfg2_balanced"].get("batch") or {}).get("value")

        if not a.no_extra_legs:
            import bench_legs
            leg("unpacked", leg_unpacked)
            leg("cfg2_trained", leg_trained)
            if not a.no_balanced and not a.unpacked:
                def leg_trainer_forest():
                    """A forest from this repo's own trainer on the bench batch (tools/bench_legs.py), and -- round 6 -- its own
                    counter passes: the trained forest and the table tune() chose go into the bench cache, `--leg
                    headline_trainer` child processes (one per counter set) evaluate the same batch on it, and the leg gets a
                    roofline of its own like the synthetic topologies."""
                    res, tf_np = bench_legs.trainer_forest(
                        rdf, frames_np, T, D, return_forest=True,
                        train_frames=np.asarray(cached(f"frames_mixed_64_{H}x{W}_at20000", lambda: synth.mixed_batch(64, first_idx=20000, h=H, w=W))))
                    if C != 4 or a.no_counters or not live:
                        return res
                    try:
                        os.makedirs(CACHE, exist_ok=True)
                        np.save(os.path.join(CACHE, f"forest_T{T}_D{D}_C{C}_trainer.npy"), tf_np)
                        run_id = os.environ.get("RDF_BENCH_RUN_ID")
                        json.dump(res["tune"], open(os.path.join(CACHE, f"tune_{run_id}_T{T}_D{D}_C{C}_trainer_{H}x{W}_F{F}.json"), "w"))
                        ft = rdf.DecisionForest.from_numpy(tf_np)
                        ft.packed(1.0)
                        assert lib.rdf_forest_set_deep_from(ft.packed(1.0).ptr, int(res["tune"]["deep_from"])) == 0
                        useful_t, st8_t = useful_lines(ft, depth)
                        alg_t = synth.algorithmic_bytes(F, H, W, 1, False, C, st8_t[0:3]) if st8_t else None
                        del ft
                        live_t = collect_counters(a, ["headline_trainer"])
                        res["roofline"] = roofline_for("headline_trainer", f"F{F}_T{T}_D{D}_C{C}_trainer", live_t, res["ms_per_step"], alg_t, useful_t)
                        res["counter_pass_seconds"] = live_t["headline_trainer"]["seconds"]
                    except Exception as e:   # noqa: BLE001 -- the leg's timing and parity stand without its counters
                        res["roofline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                    return res
                leg("cfg2_trainer_forest", leg_trainer_forest)
            leg("hand_pipeline", lambda: bench_legs.hand_pipeline(rdf))
            leg("mean_shift", lambda: bench_legs.mean_shift(rdf))
            leg("train", lambda: bench_legs.train(rdf))
            leg("train_256_frames_d16", lambda: bench_legs.train(rdf, images=256, depth=16, proposals=1024, check=False))

        # ---- config 5's shard last: it frees the headline's buffers first (2 GiB forest + 2.5 GiB packed tables) ----
        if not a.no_cfg5:
            del forest, depth, labels, scratch, sharded, lf
            torch.cuda.empty_cache()
            c5 = leg_cfg5(5, 2, True)
            c5["roofline"] = roofline_for("cfg5", f"F{a.cfg5_frames}_T{a.cfg5_trees}_D{a.cfg5_depth}_C4_full_1280x720", live, c5["kernel_ms"],
                                          c5.pop("algorithmic_bytes", None), c5.pop("useful_lines", None))
            out["cfg5_shard"] = c5
            if not a.no_balanced and not a.unpacked:
                try:
                    c5b = leg_cfg5(5, 2, True, "balanced")
                    c5b["roofline"] = roofline_for("cfg5_balanced", f"F{a.cfg5_frames}_T{a.cfg5_trees}_D{a.cfg5_depth}_C4_balanced_1280x720", live,
                                                   c5b["kernel_ms"], c5b.pop("algorithmic_bytes", None), c5b.pop("useful_lines", None))
                except Exception as e:   # noqa: BLE001 -- the headline is measured; it must still be printed
                    c5b = {"error": f"{type(e).__name__}: {e}"[:400]}
                out["cfg5_balanced"] = c5b
        if live:
            out["counter_passes"] = {leg: {"seconds": v["seconds"], "log": v["log"]} for leg, v in live.items()}

    if rank == 0:
        if multi:
            out["distributed"]["total_seconds"] = round(time.perf_counter() - t_main, 1)    # this process, start to line
        emit(out, a)
    if multi:
        dist.barrier()
        for ring in (pg, pg_d):                 # unmap on the peers before rank 0 frees the buffer
            if ring is not None and ring.ok:        # (`ok` is agreed on every rank: the barrier below is entered by all or none)
                if rank != 0:
                    ring.close()
                dist.barrier()
                if rank == 0:
                    ring.close()
        for h in handles:
            if h is not None:
                lib.rdf_stream_destroy(h)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
