#!/bin/bash
# Individual, time-boxed PMC passes (TCP/TA counters crashed rocprofv3 when grouped; see profiles/).
# usage: tools/pmc2.sh <out> "<bench args, include --headline-only or --leg X: a full bench.py run profiles itself>" pass1:"C1 C2" pass2:"C3" ...
set -u
OUT=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
for spec in "$@"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  echo "pass $name: $ctrs" | tee -a $R/gpurun_out/$OUT/log.txt
  timeout -k 5 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/$OUT/$name -- python3 $R/bench.py $ARGS \
      > $R/gpurun_out/$OUT/$name.json 2> $R/gpurun_out/$OUT/$name.err
  echo "  rc=$?" | tee -a $R/gpurun_out/$OUT/log.txt
done
cd $R && python3 tools/pmc_summary.py gpurun_out/$OUT > gpurun_out/$OUT/summary.txt 2>&1
grep -A40 "k_eval_forest<512, true, 4, false\|k_eval_forest<1024, true, 4, false\|k_eval_forest<256, true, 4, false" gpurun_out/$OUT/summary.txt | head -80
