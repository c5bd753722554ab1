#!/bin/bash
# L2-side request mix of the solve kernel: usage: bash tools/collect_pmc5.sh <tag>
set -u
TAG=${1:-pmc5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout 150 rocprofv3 --pmc "$@" -d $OUT/$name -o r -- python3 tools/perf_probe.py 131 512 0.4 smooth 256 > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/rc.txt
}
run a TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
run b TCC_HIT_sum TCC_MISS_sum WRITE_SIZE
run c FETCH_SIZE
run d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TA_BUSY_avr SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run e TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
run f TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
for n in a b c d e f; do
  db=$(find $OUT/$n -name "*.db" | head -1)
  echo "== $n"; grep "solves/s" $OUT/$n.log | cut -c1-120
  [ -n "$db" ] && python3 tools/rocpd_pmc.py $db | grep "k_fim" | head -12
done > $OUT/summary.txt 2>&1
cat $OUT/rc.txt $OUT/summary.txt
