#!/usr/bin/env python3
"""bench.py's FULL result (bench_full.json / --full-json, or the same object from its stderr) -> profiles/<tag>_bench.json
(pretty-printed), the compact stdout line -> profiles/<tag>_bench_line.json (as printed: what the driver parses), and
profiles/<tag>_roofline_counters.json (the counters of its --pmc legs: what bench.py falls back to when a run cannot collect
counters itself).
usage: tools/save_profile.py gpurun_out/<dir>/bench_full.json <tag> [gpurun_out/<dir>/bench.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(path, tag, line_path=None):
    line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert "roofline" in d and "levels" in d["roofline"], "this is the compact line: pass the full result (bench_full.json)"
    json.dump(d, open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w"), indent=1)
    if line_path:
        compact = [l for l in open(line_path).read().splitlines() if l.startswith("{")][-1]
        assert len(compact) < 4096, len(compact)
        json.loads(compact)
        open(os.path.join(ROOT, "profiles", f"{tag}_bench_line.json"), "w").write(compact + "\n")
    c = d["config"]
    keys = {"roofline": f"F{c['frames_per_gpu']}_T{c['trees']}_D{c['tree_depth']}_C{c['classes']}_{c['topology']}"}
    out = {}
    for leg, r in (("headline", d.get("roofline")), ("cfg2", d.get("cfg2_single_frame", {}).get("roofline")),
                   ("cfg5", d.get("cfg5_shard", {}).get("roofline")),
                   ("headline_balanced", (d.get("cfg2_balanced", {}).get("batch") or {}).get("roofline")),
                   ("cfg5_balanced", d.get("cfg5_balanced", {}).get("roofline")),
                   ("headline_trainer", d.get("cfg2_trainer_forest", {}).get("roofline"))):
        if not r or not r.get("counters") or "child passes" not in (r.get("counters_source") or ""):
            continue
        key = {"headline": keys["roofline"], "cfg2": keys["roofline"].replace(f"F{c['frames_per_gpu']}_", "F1_"),
               "cfg5": "F32_T8_D22_C4_full_1280x720",
               "headline_balanced": keys["roofline"].replace(f"_{c['topology']}", "_balanced"),
               "cfg5_balanced": "F32_T8_D22_C4_balanced_1280x720",
               "headline_trainer": keys["roofline"].replace(f"_{c['topology']}", "_trainer")}[leg]
        out[key] = {"kernel": r["kernel"], "counters": r["counters"], "kernel_ms_of_that_run": r["kernel_ms"],
                    "collected_by": f"bench.py's rocprofv3 --pmc child passes ({tag})"}
    if out:
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag.split('_')[0]}_roofline_counters.json"), "w"), indent=1)
    print(f"saved profiles/{tag}_bench.json" + (f", profiles/{tag.split('_')[0]}_roofline_counters.json" if out else ""))


if __name__ == "__main__":
    main(*sys.argv[1:4])
