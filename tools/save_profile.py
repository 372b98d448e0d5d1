#!/usr/bin/env python3
"""bench.py's JSON line -> profiles/<tag>_bench.json (the line, pretty-printed) and profiles/r04_roofline_counters.json
(the counters of its three --pmc legs: what bench.py falls back to when a run cannot collect counters itself).
usage: tools/save_profile.py gpurun_out/<dir>/bench.json <tag>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(path, tag):
    line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    json.dump(d, open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w"), indent=1)
    c = d["config"]
    keys = {"roofline": f"F{c['frames_per_gpu']}_T{c['trees']}_D{c['tree_depth']}_C{c['classes']}_{c['topology']}"}
    out = {}
    for leg, r in (("headline", d.get("roofline")), ("cfg2", d.get("cfg2_single_frame", {}).get("roofline")),
                   ("cfg5", d.get("cfg5_shard", {}).get("roofline")),
                   ("headline_balanced", (d.get("cfg2_balanced", {}).get("batch") or {}).get("roofline")),
                   ("cfg5_balanced", d.get("cfg5_balanced", {}).get("roofline"))):
        if not r or not r.get("counters") or "child passes" not in (r.get("counters_source") or ""):
            continue
        key = {"headline": keys["roofline"], "cfg2": keys["roofline"].replace(f"F{c['frames_per_gpu']}_", "F1_"),
               "cfg5": "F32_T8_D22_C4_full_1280x720",
               "headline_balanced": keys["roofline"].replace(f"_{c['topology']}", "_balanced"),
               "cfg5_balanced": "F32_T8_D22_C4_balanced_1280x720"}[leg]
        out[key] = {"kernel": r["kernel"], "counters": r["counters"], "kernel_ms_of_that_run": r["kernel_ms"],
                    "collected_by": f"bench.py's rocprofv3 --pmc child passes ({tag})"}
    if out:
        json.dump(out, open(os.path.join(ROOT, "profiles", "r04_roofline_counters.json"), "w"), indent=1)
    print(f"saved profiles/{tag}_bench.json" + (", profiles/r04_roofline_counters.json" if out else ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
