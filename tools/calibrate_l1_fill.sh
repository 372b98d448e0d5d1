#!/bin/bash
# tools/ubench_l1_fill on the GPU box: the timing run, then two counter passes that say what share of each case's L1
# line accesses really were fills (the "L1-resident" lines of a mixed case can be evicted by the fills next to them).
# usage (through gpurun): bash tools/calibrate_l1_fill.sh   -> gpurun_out/l1_fill/{timing.txt,fill_share.txt}
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/l1_fill
mkdir -p $O
$R/tools/bin/ubench_l1_fill > $O/timing.txt 2>&1 || exit 1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_fill -- $R/tools/bin/ubench_l1_fill > /dev/null 2> $O/pmc_fill.err || exit 1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/pmc_acc -- $R/tools/bin/ubench_l1_fill > /dev/null 2> $O/pmc_acc.err || exit 1
python3 - "$O" > $O/fill_share.txt <<'PY'
import csv, glob, sys
o = sys.argv[1]
def read(d, name):
    rows = []
    for f in glob.glob(f"{o}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "k_lines" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(rows)
fill, acc = read("pmc_fill", "TCP_TCC_READ_REQ_sum"), read("pmc_acc", "TCP_TOTAL_CACHE_ACCESSES_sum")
print("dispatch kernel fills accesses fill_share   (3 dispatches per case, in the order of timing.txt)")
for (i, k, f), (_, _, a) in zip(fill, acc):
    print(i, k.split("(")[0][-12:], int(f), int(a), round(f / a, 4))
PY
cat $O/timing.txt
cat $O/fill_share.txt
