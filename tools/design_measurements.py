#!/usr/bin/env python3
"""DESIGN.md section 5's round-6 table, generated from the committed profiles so that every figure in it IS a figure of a
file under profiles/ (the review of round 4 found the design document quoting other numbers than the files it cited).

    python3 tools/design_measurements.py            prints the block
    python3 tools/design_measurements.py --write    replaces the block between the markers in DESIGN.md

tests/test_docs.py checks that DESIGN.md carries exactly this block."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- r06-measurements:begin (tools/design_measurements.py) -->", "<!-- r06-measurements:end -->"
TAG = "r06"


def kernel_stats(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    rows = [r for r in csv.DictReader(open(path)) if "k_eval_forest" in r.get("Name", "")]
    if not rows:
        return None
    r = max(rows, key=lambda x: float(x["TotalDurationNs"]))
    short = r["Name"].split("k_eval_forest", 1)[1].split(">", 1)[0] + ">"
    return {"kernel": "k_eval_forest" + short.replace(", ", ","), "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
            "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}


def block():
    d = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_bench.json")))
    r = d["roofline"]
    lv = r["levels"]
    ta = lv["l1_ta"]
    b = d["cfg2_balanced"]
    br = b["batch"]["roofline"]
    c5, c5b = d["cfg5_shard"], d["cfg5_balanced"]
    c5r, c5br = c5["roofline"], c5b["roofline"]
    c2 = d["cfg2_single_frame"]
    cb = d["cpu_baseline"]
    rows = []

    def row(q, v, src):
        rows.append(f"| {q} | {v} | {src} |")

    row("batch throughput, 1 GPU, \"full\" topology (the metric's config)",
        f"**{d['value']} Mpix/s**, median step {d['ms_per_step']} ms (mean over the bracketed steps {d['ms_per_step_mean']} ms = {d['value_mean']} Mpix/s); "
        f"per evaluated pixel ({d['valid_pixel_share']:.0%} of the batch) {d['value_valid_pixels']} Mpix/s; deep-level table tuned: {d['config']['deep_level_table']['deep_from']} (heap-order records); "
        f"round 5's driver run: 13 323 Mpix/s, 3.911 ms",
        "`value`, `ms_per_step`")
    ks = kernel_stats(f"{TAG}_kernel_stats.csv")
    if ks:
        tr = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_bench_under_rocprof.json")))
        row("the same kernel under `rocprofv3 --kernel-trace --stats` (one traced run: hipEvents beside the trace)",
            f"{ks['avg_ms']:.3f} ms average over {ks['calls']} launches (min {ks['min_ms']:.3f}, max {ks['max_ms']:.3f}) of `{ks['kernel']}`; hipEvent median of that run {tr['ms_per_step']} ms",
            f"`profiles/{TAG}_kernel_stats.csv`, `profiles/{TAG}_bench_under_rocprof.json`")
    row("roofline, headline kernel",
        f"bound **{r['bound']}**: `frac` {r['frac']} ({ta['l1_line_accesses_per_launch'] / 1e9:.3f}·10⁹ L1 line accesses × {ta['cycles_per_line_access']} cycles at a fill share of {ta['fill_share']}), "
        f"`useful_frac` {r['useful_frac']}; `TA_TA_BUSY` {ta['ta_busy_frac_counter']} (model {ta['ta_busy_model']}); VALU {lv['valu']['frac']}, L2→L1 {lv['l2_l1']['frac']}, "
        f"fabric side {lv['hbm']['achieved']} GB/s = {lv['hbm']['frac']} of 8 TB/s ({r['traffic'] / 1e9:.2f} GB per launch against {r['algorithmic']['bytes_per_launch'] / 1e9:.1f} GB algorithmic: L2 hit rate {r['l2_hit_rate']}); "
        f"SURVEY 8(d)'s formula as written (`survey_8d_frac`): {r['algorithmic']['over_hbm_peak']} — above 1, a fraction of nothing",
        "`roofline`")
    row("**the same batch on a forest whose deep levels are occupied** (balanced T4/D20)",
        f"**{b['batch']['value']} Mpix/s, {b['batch']['ms_per_step']} ms** (round 5: 7 729 Mpix/s, 6.74 ms; round 4: 7.70–7.91 ms; heap-order records: 11.2 ms); tuned: deep blocks from level {b['tune']['deep_from']} "
        f"(sample: {b['tune']['tried']}); one dense frame {round(b['kernel_ms'] * 1e3, 1)} µs (round 5: 124); {b['parity']['frames_checked']} frames against the oracle: {b['parity']['differing_pixels']} differing pixels",
        "`cfg2_balanced`, `value_balanced`")
    ksb = kernel_stats(f"{TAG}_kernel_stats_balanced.csv")
    if ksb:
        trb = json.load(open(os.path.join(ROOT, "profiles", f"{TAG}_bench_under_rocprof_balanced.json")))
        row(f"its kernel under `rocprofv3 --kernel-trace --stats` (`--topology balanced --deep-from {b['tune']['deep_from']}`)",
            f"{ksb['avg_ms']:.3f} ms average over {ksb['calls']} launches (min {ksb['min_ms']:.3f}, max {ksb['max_ms']:.3f}) of `{ksb['kernel']}`; hipEvent median of that run {trb['ms_per_step']} ms",
            f"`profiles/{TAG}_kernel_stats_balanced.csv`, `profiles/{TAG}_bench_under_rocprof_balanced.json`")
    bh, bt = br["levels"]["hbm"], br["levels"]["l1_ta"]
    row("its roofline",
        f"fabric side **{bh['bytes_per_launch'] / 1e9:.1f} GB per launch = {bh['achieved']} GB/s = {bh['frac']} of 8 TB/s, {bh['frac_of_gather_ceiling']} of the {bh['gather_ceiling']:.0f} GB/s a pure gather of random lines reaches**; "
        f"L1 level {bt['frac']} ({bt['l1_line_accesses_per_launch'] / 1e9:.2f}·10⁹ accesses; round 4's seven-load walk issued 3.50·10⁹), `TA_TA_BUSY` {bt['ta_busy_frac_counter']} (model {bt['ta_busy_model']}), "
        f"VALU {br['levels']['valu']['frac']}, L2 hit rate {br['l2_hit_rate']}; `bound` {br['bound']} ({br['frac']})",
        "`cfg2_balanced.batch.roofline`")
    row("config 5's shard (32 dense 1280×720 frames, T8/D22/C4), \"full\" topology",
        f"{c5['value']} Mpix/s, {c5['kernel_ms']} ms; tuned: {c5['tune']['deep_from']} ({c5['tune']['tried']}); bound {c5r['bound']} {c5r['frac']}, fabric side {c5r['levels']['hbm']['frac']}; "
        f"2 frames against the oracle: {c5['parity']['differing_pixels']} differing pixels", "`cfg5_shard`")
    ch = c5br["levels"]["hbm"]
    row("**config 5's shard, balanced T8/D22** (3.6 GiB of packed tables: HBM-resident — the roofline point BASELINE configs[4] names)",
        f"**{c5b['value']} Mpix/s, {c5b['kernel_ms']} ms** (round 5: 1 643 Mpix/s, 17.94 ms — 1280 = 20 × 64 columns: no narrow tiles here; round 4: 22.0–23.1 ms; heap-order records: 38.4 ms); tuned: blocks from level {c5b['tune']['deep_from']}; "
        f"**{ch['bytes_per_launch'] / 1e9:.1f} GB per launch = {ch['achieved']} GB/s = {ch['frac']} of the 8 TB/s peak = {ch['frac_of_gather_ceiling']} of the measured gather ceiling**; "
        f"`TA_TA_BUSY` {c5br['levels']['l1_ta']['ta_busy_frac_counter']} (model {c5br['levels']['l1_ta']['ta_busy_model']}), L1 level {c5br['levels']['l1_ta']['frac']}, L2 hit rate {c5br['l2_hit_rate']}; "
        f"2 frames against the oracle: {c5b['parity']['differing_pixels']} differing pixels", "`cfg5_balanced`")
    row("config 2, one dense frame per launch", f"{round(c2['kernel_ms'] * 1e3, 1)} µs ({c2['value']} Mpix/s; round 5: 77.5); L1 level {c2['roofline']['levels']['l1_ta']['frac']}, VALU {c2['roofline']['levels']['valu']['frac']}; "
        f"trained-like topology {round(d['cfg2_trained']['kernel_ms'] * 1e3, 1)} µs", "`cfg2_single_frame`, `cfg2_trained`")
    row("config 3, `LayeredDecisionForest.run`, 2 layers, r = 2, one live-like frame", f"{round(d['cfg3_layered_run']['ms_per_frame_wall'] * 1e3, 1)} µs per frame", "`cfg3_layered_run`")
    tf = d["cfg2_trainer_forest"]
    tfr = tf.get("roofline") or {}
    tfl = tfr.get("levels") or {}
    row("a forest from the repo's own trainer (T4/D20) on the bench batch",
        f"{tf['value']} Mpix/s, {tf['ms_per_step']} ms; tuned: {tf['tune']['deep_from']} ({tf['tune']['tried']}); its own counter passes (round 6): bound {tfr.get('bound')} {tfr.get('frac')}, "
        f"VALU {(tfl.get('valu') or {}).get('frac')}, L2→L1 {(tfl.get('l2_l1') or {}).get('frac')}, fabric side {(tfl.get('hbm') or {}).get('frac')} of 8 TB/s, L2 hit rate {tfr.get('l2_hit_rate')} — no level near its limit: "
        f"between the cache-resident \"full\" topology and the occupied one ({tf['share_of_level_visited'][-1]:.0%} of level 19 visited, a 24-MB working set of hot records), it pays the walk's dependent chain with L2 latencies in it", "`cfg2_trainer_forest`")
    row("frames from and labels to HOST memory", f"serial {d['pcie_inclusive']['value']} Mpix/s; `HostFramesEvaluator` {d['pcie_inclusive_pipelined']['value']} Mpix/s ({d['pcie_inclusive_pipelined']['ms_per_step']} ms per step)",
        "`pcie_inclusive`, `pcie_inclusive_pipelined`")
    row("other legs", f"reference-layout forest {d['unpacked']['value']} Mpix/s; trained-like batch {d['cfg2_trained']['batch']['value']} Mpix/s; per-hand chain as one hipGraph {d['hand_pipeline']['us_per_hand_per_frame_as_hipgraph']} µs; "
        f"training 64 frames D12 {d['train']['seconds']} s, 256 frames D16 {d['train_256_frames_d16']['seconds']} s", "`unpacked`, `cfg2_trained`, `hand_pipeline`, `train*`")
    row("CPU baseline (this repo's C restatement, OpenMP)", f"{cb['value']} Mpix/s on {cb['cores']} cores; {cb['parity_frames']} frames of the timed step checked: {cb['differing_pixels']} differing pixels; GPU / CPU = {d['value'] / cb['value']:.0f}×; "
        f"numpy restatement, 1 core: {d['cpu_baseline_numpy']['value']} Mpix/s", "`cpu_baseline`, `cpu_baseline_numpy`")
    head = [f"**Round 6** (`profiles/{TAG}_bench.json` = the full result of ONE default `python bench.py` run on the final commit; `profiles/{TAG}_bench_line.json` = its compact stdout line as the driver parses it; "
            "boxes of the pool differ by about 1.5 %, so another run's figures move by that much):", "",
            "| quantity | value | source (key of the full result) |", "|---|---|---|"]
    return "\n".join([BEGIN] + head + rows + [END])


if __name__ == "__main__":
    text = block()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        if BEGIN in s:
            s = s[:s.index(BEGIN)] + text + s[s.index(END) + len(END):]
        else:
            s = s.replace("@@ROUND6_MEASUREMENT@@", text)
        open(p, "w").write(s)
    else:
        print(text)
