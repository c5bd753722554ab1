#!/bin/bash
# SQ-side picture of the solve kernel (VALU occupancy, wait cycles): bash tools/collect_pmc6.sh <tag> [lib]
set -u
TAG=${1:-pmc6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "${2:-}" ] && export DSA_LIB_PATH=$2
run() {
  local name=$1; shift
  timeout 150 rocprofv3 --pmc "$@" -d $OUT/$name -o r -- python3 tools/perf_probe.py 131 1024 1.25 smooth 256 > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/rc.txt
}
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run c SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
run d SQ_WAVES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run e SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
for n in a b c d e; do
  db=$(find $OUT/$n -name "*.db" | head -1)
  echo "== $n"; grep "solves/s" $OUT/$n.log | cut -c1-160
  [ -n "$db" ] && python3 tools/rocpd_pmc.py $db | grep "k_fim" | head -12
done > $OUT/summary.txt 2>&1
cat $OUT/rc.txt $OUT/summary.txt
