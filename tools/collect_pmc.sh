#!/bin/bash
# PMC passes for the coarse fixed-point kernel; one counter set per run (rocprofv3 --pmc, no traces).
# usage (on the GPU box, from the repo root): bash tools/collect_pmc.sh <tag>
set -u
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $OUT/$name -o r -- python3 tools/perf_probe.py 131 256 3 smooth 256 > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
for n in sq1 sq2 tcc1 tcc2 tcp; do
  db=$(find $OUT/$n -name "*.db" | head -1)
  echo "== $n ($db)"; tail -3 $OUT/$n.log
  [ -n "$db" ] && python3 tools/rocpd_pmc.py $db | grep -v "^#" | head -40
done > $OUT/summary.txt 2>&1
