#!/bin/bash
# Kernel-trace stats of the headline workload (run through gpurun): rocprofv3's own average duration of the forest kernel
# next to bench.py's hipEvent figure from the same traced run.  The counters (--pmc) are collected by bench.py itself.
# usage: tools/profile.sh <tag> [bench args]    -> gpurun_out/<tag>/{kernel_stats.csv, bench_traced.json}
# Pass the forest's deep-level table explicitly (--deep-from 0 for the headline's "full" topology, --deep-from 15 with
# --topology balanced: what DecisionForest.tune picks) so that the trace holds no tuning launches of the same kernel.
set -u
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only "$@" \
    > $O/bench_traced.json 2> $O/trace.err
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
head -6 $O/kernel_stats.csv
cat $O/bench_traced.json
