#!/bin/bash
# Kernel-trace stats + HBM traffic counters for the bench workload (run through gpurun).
# usage: tools/profile.sh <tag>     -> gpurun_out/<tag>/{kernel_stats.csv, fetch/, write/, summary.txt}
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only \
    > $O/bench_traced.json 2> $O/trace.err
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
for spec in fetch:"FETCH_SIZE TCC_HIT_sum" write:"WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" grbm:"GRBM_GUI_ACTIVE" \
            tcp:"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
            sq:"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  timeout -k 5 150 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -- python3 $R/bench.py --steps 3 --warmup 1 --headline-only \
      > $O/$name.json 2> $O/$name.err
  echo "pass $name rc=$?"
done
cd $R && python3 tools/pmc_summary.py gpurun_out/$TAG > $O/summary.txt 2>&1
head -12 $O/kernel_stats.csv
grep -A16 "k_eval_forest<256, true, 4, false, true, 4, false>" $O/summary.txt | head -20
python3 tools/make_traffic_json.py $O/summary.txt "F128_T4_D20_C4_full" "k_eval_forest<256, true, 4, false, true, 4, false>" > $O/roofline_traffic.json
cat $O/roofline_traffic.json
