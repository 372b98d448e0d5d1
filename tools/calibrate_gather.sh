#!/bin/bash
# The memory side's ceiling for the deep-block walk's access pattern (tools/ubench_gather.hip), and what the counters of
# tools/roofline.py read at that ceiling (TA_TA_BUSY: a texture addresser that waits for the fabric counts as busy).
# usage (through gpurun): tools/calibrate_gather.sh <tag>  -> gpurun_out/<tag>/{ubench_gather.txt, gather_<shape>_<MB>.txt}
set -u
TAG=${1:-gather}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 $R/tools/bin/ubench_gather 400 > $O/ubench_gather.txt 2>&1; echo "ubench_gather rc=$?"
cat $O/ubench_gather.txt
for shape in coop16:146 coop16:1200 seven24:1200; do
  s=${shape%%:*}; mb=${shape#*:}
  for spec in ta:"TA_TA_BUSY_sum GRBM_GUI_ACTIVE" tcp:"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" tcc:"TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    name=${spec%%:*}; ctrs=${spec#*:}
    timeout -k 5 120 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc_${s}_${mb}_$name -- $R/tools/bin/ubench_gather 400 $s $mb > $O/pmc_${s}_${mb}_$name.out 2> $O/pmc_${s}_${mb}_$name.err
    echo "pass $s $mb $name rc=$?"
  done
done
cd $R && python3 - <<'PY' $O
import csv, glob, os, sys
o = sys.argv[1]
for d in sorted(glob.glob(os.path.join(o, "pmc_*_ta"))):
    key = os.path.basename(d)[4:-3]
    vals, ns = {}, {}
    for f in glob.glob(os.path.join(o, f"pmc_{key}_*", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_gather" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in rows)            # the last launch: a timed one
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                vals[r["Counter_Name"]] = float(r["Counter_Value"])
                ns[r["Counter_Name"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    t = ns.get("TA_TA_BUSY_sum", 0) * 1e-9
    clk = vals.get("GRBM_GUI_ACTIVE", 0) / 8 / t if t else 0
    acc = vals.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)
    line = (f"{key:14s} launch {t * 1e3:7.3f} ms  clock {clk / 1e9:5.2f} GHz  TA busy {vals.get('TA_TA_BUSY_sum', 0) / 256 / (t * clk) if t else 0:5.3f}  "
            f"L1 line accesses {acc:.4g} = {acc / 256 / (t * clk) if t else 0:5.3f} per CU and cycle  fills {vals.get('TCP_TCC_READ_REQ_sum', 0):.4g}  "
            f"fabric requests {vals.get('TCC_EA0_RDREQ_sum', 0):.4g} = {vals.get('TCC_EA0_RDREQ_sum', 0) * 128 / ns.get('TCC_EA0_RDREQ_sum', 1):7.1f} GB/s in that pass  "
            f"L2 hits {vals.get('TCC_HIT_sum', 0):.4g} misses {vals.get('TCC_MISS_sum', 0):.4g}")
    print(line)
    open(os.path.join(o, "gather_counters.txt"), "a").write(line + "\n")
PY
