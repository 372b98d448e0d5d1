#!/bin/bash
# What TA_TA_BUSY charges per L2 miss BELOW the fabric's ceiling (where the forest kernel runs): tools/ubench_gather.hip with a
# dependent integer hash between two fetches of a wave ("think"), for the cooperative LDS-DMA fetch and for one divergent
# 16-byte load per lane, on a 1.2-GB table (every fill misses the L2).  Per point: GB/s, TA busy, L1 accesses, and
#   busy cycles per miss = (TA_TA_BUSY / CUs - L1 accesses x cycles per access at that fill share / CUs) / (misses / CUs)
# -- the term tools/roofline.py adds in ta_busy_model.
# usage (through gpurun): tools/calibrate_ta_busy.sh <tag>  -> gpurun_out/<tag>/ta_busy_per_miss.txt
set -u
TAG=${1:-tabusy}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in coop16 one24; do
  for think in 0 100 200 400 800; do
    for spec in ta:"TA_TA_BUSY_sum GRBM_GUI_ACTIVE" tcp:"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" tcc:"TCC_EA0_RDREQ_sum TCC_MISS_sum"; do
      name=${spec%%:*}; ctrs=${spec#*:}
      timeout -k 5 120 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc_${shape}_${think}_$name -- $R/tools/bin/ubench_gather 400 $shape 1200 $think > $O/pmc_${shape}_${think}_$name.out 2> $O/pmc_${shape}_${think}_$name.err
      echo "pass $shape think $think $name rc=$?"
    done
  done
done
cd $R && python3 - <<'PY' $O
import csv, glob, os, sys
sys.path.insert(0, "tools")
import roofline
o = sys.argv[1]
out = open(os.path.join(o, "ta_busy_per_miss.txt"), "w")
for shape in ("coop16", "one24"):
    for think in (0, 100, 200, 400, 800):
        vals, ns = {}, {}
        for f in glob.glob(os.path.join(o, f"pmc_{shape}_{think}_*", "**", "*counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_gather" in r["Kernel_Name"]]
            if not rows:
                continue
            last = max(int(r["Dispatch_Id"]) for r in rows)
            for r in rows:
                if int(r["Dispatch_Id"]) == last:
                    vals[r["Counter_Name"]] = float(r["Counter_Value"])
                    ns[r["Counter_Name"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if "TA_TA_BUSY_sum" not in vals or "TCC_MISS_sum" not in vals or "TCP_TOTAL_CACHE_ACCESSES_sum" not in vals:
            continue
        t = ns["TA_TA_BUSY_sum"] * 1e-9
        clk = vals["GRBM_GUI_ACTIVE"] / 8 / t
        cyc = t * clk
        acc, fills, miss = vals["TCP_TOTAL_CACHE_ACCESSES_sum"], vals["TCP_TCC_READ_REQ_sum"], vals["TCC_MISS_sum"]
        share = min(1.0, fills / acc)
        l1 = acc * roofline.l1_cycles_per_access(share) / 256
        busy = vals["TA_TA_BUSY_sum"] / 256
        gbs = vals["TCC_EA0_RDREQ_sum"] * 128 / ns["TCC_EA0_RDREQ_sum"]
        line = (f"{shape:7s} think {think:4d}: {gbs:7.0f} GB/s = {gbs / roofline.GATHER_CEILING_GBS:5.2f} of the ceiling  launch {t * 1e3:7.3f} ms  TA busy {busy / cyc:5.3f}  "
                f"L1 level {l1 / cyc:5.3f} (fill share {share:4.2f})  busy cycles per L2 miss {(busy - l1) / (miss / 256):5.2f}")
        print(line)
        out.write(line + "\n")
PY
