#!/bin/bash
# Microbenchmarks that calibrate the roofline model of the forest kernel (run through gpurun; binaries are built in
# the authoring container: hipcc -O3 --offload-arch=gfx950 -o tools/bin/<name> tools/<name>.hip).
#   ubench_valu   cycles a wave64 VALU instruction holds a SIMD's issue port at 1/2/4/5/8 waves per SIMD
#   ubench_fetch  what FETCH_SIZE / TCC_EA0_RDREQ* / TCC_MISS / TCP_TCC_READ_REQ report for a known number of line fills
# usage: tools/calibrate.sh <tag>   -> gpurun_out/<tag>/{ubench_valu.txt, ubench_fetch_time.txt, fetch_*/, fetch_summary.txt}
set -u
TAG=${1:-calib}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 $R/tools/bin/ubench_valu 20000 > $O/ubench_valu.txt 2>&1; echo "ubench_valu rc=$?"
timeout -k 10 120 $R/tools/bin/ubench_fetch > $O/ubench_fetch_time.txt 2>&1; echo "ubench_fetch rc=$?"
rocprofv3 -L > $O/counters_list.txt 2>&1
for spec in fetch:"FETCH_SIZE" write:"WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" ea:"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum" \
            tcp:"TCP_TCC_READ_REQ_sum" tcpacc:"TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  timeout -k 5 150 rocprofv3 --pmc $ctrs --output-format csv -d $O/fetch_$name -- $R/tools/bin/ubench_fetch > $O/fetch_$name.out 2> $O/fetch_$name.err
  echo "pass $name rc=$?"
done
cd $R && python3 tools/pmc_summary.py gpurun_out/$TAG > $O/fetch_summary.txt 2>&1
cat $O/ubench_valu.txt $O/ubench_fetch_time.txt
grep -v "k_fill" $O/fetch_summary.txt | head -60
