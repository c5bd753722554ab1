#!/bin/bash
# third PMC set: is the solve kernel issue-bound at full occupancy?  usage: bash tools/collect_pmc3.sh <tag>
set -u
TAG=${1:-pmc3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $OUT/$name -o r -- python3 tools/perf_probe.py 131 1024 0.4 smooth 256 > $OUT/$name.log 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run b SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU
run c GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT
for n in a b c; do
  db=$(find $OUT/$n -name "*.db" | head -1)
  echo "== $n"; grep "solves/s" $OUT/$n.log | cut -c1-120
  [ -n "$db" ] && python3 tools/rocpd_pmc.py $db | grep "k_fim" | head -12
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
