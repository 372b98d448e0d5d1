#!/bin/bash
# PMC passes over `python3 bench.py` on the GPU box (run through gpurun).  Each pass is its own
# rocprofv3 run with --pmc only (never combined with tracing), per MI355X_MICROARCH.md.
# usage: tools/pmc.sh <out_dir_under_gpurun_out> [bench args...]
set -u
OUT=${1:-pmc}; shift || true
ARGS=${@:---steps 3 --warmup 1 --headline-only}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
pass() { # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$name -- python3 $R/bench.py $ARGS \
      > $R/gpurun_out/$OUT/$name.json 2> $R/gpurun_out/$OUT/$name.err || echo "pass $name failed rc=$?"
}
pass fetch  FETCH_SIZE TCC_HIT_sum
pass write  WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pass sqcyc  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass sqinst SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass ta     TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp    TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcp2   TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
pass grbm   GRBM_GUI_ACTIVE GRBM_COUNT
cd $R && python3 tools/pmc_summary.py gpurun_out/$OUT > gpurun_out/$OUT/summary.txt 2>&1
cat gpurun_out/$OUT/summary.txt
